"""Multi-GPU helpers: one process per GPU, torch.distributed ("nccl" = RCCL over xGMI on ROCm).

Only the naturally sharded axis is partitioned: the batch. Sampling shards need no collective inside
the loop (each rank denoises its own samples); training averages gradients with one all-reduce of
the flattened gradient buffer per step. Every function also works on CPU tensors with the gloo
backend, which is how the logic is tested without GPUs.
"""
import torch
import torch.distributed as dist


def shard_range(total, world_size, rank):
    """Contiguous [start, stop) of `total` items for `rank`; shard sizes differ by at most one."""
    base, extra = divmod(int(total), int(world_size))
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def _world(group=None):
    if not (dist.is_available() and dist.is_initialized()):
        return 1, 0
    return dist.get_world_size(group), dist.get_rank(group)


def _host_staged(group=None):
    """gloo moves host memory: device tensors are staged through the host for it (rehearsals of the
    multi-process path on one GPU; RCCL takes device pointers directly)."""
    return dist.get_backend(group) == "gloo"


def all_reduce_(t, op=None, group=None):
    """In-place all-reduce of `t` on any backend."""
    op = dist.ReduceOp.SUM if op is None else op
    if t.is_cuda and _host_staged(group):
        h = t.cpu()
        dist.all_reduce(h, op=op, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=op, group=group)
    return t


def broadcast_parameters(module, src=0, group=None):
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    for t in list(module.parameters()) + list(module.buffers()):
        if t.is_cuda and _host_staged(group):
            h = t.data.cpu()
            dist.broadcast(h, src=src, group=group)
            t.data.copy_(h)
        else:
            dist.broadcast(t.data, src=src, group=group)


def average_gradients(params, group=None):
    """All-reduce (mean) of the gradients of `params`: one collective over the flattened buffer
    (4 bytes per parameter: 31-50 MB for the three UNets). Afterwards every `.grad` is a view into
    that buffer, so nothing is copied back (one concatenation, one collective, one scale per step)."""
    if not (dist.is_available() and dist.is_initialized()):
        return
    world, _ = _world(group)
    with_grad = [p for p in params if p.grad is not None]
    if not with_grad:
        return
    flat = torch.cat([p.grad.reshape(-1) for p in with_grad])
    all_reduce_(flat, group=group)
    flat.div_(world)
    off = 0
    for p in with_grad:
        n = p.grad.numel()
        p.grad = flat[off:off + n].view_as(p.grad)
        off += n


def sample_sharded(diffusion, x_shape, device, gather=True, noise=None, group=None, **sample_kwargs):
    """Run diffusion.sample on this rank's slice of the batch axis.

    noise: optional (x_T, [z ...]) for the WHOLE batch; each rank injects its slice, so that the
    sharded result equals the single-process one sample for sample.
    Returns the reference's stage list; with gather=True every rank gets the whole batch
    (concatenated in rank order), otherwise only its own slice.
    """
    world, rank = _world(group)
    lo, hi = shard_range(x_shape[0], world, rank)
    local_shape = (hi - lo,) + tuple(x_shape[1:])
    if noise is not None:
        sample_kwargs["noise"] = (noise[0][lo:hi], [z[lo:hi] for z in noise[1]])
    stages = diffusion.sample(local_shape, device, **sample_kwargs) if hi > lo else []
    if world == 1 or not gather:
        return stages
    gathered = [None] * world
    dist.all_gather_object(gathered, [(x, t) for x, t in stages], group=group)
    ref = next(g for g in gathered if g)
    out = []
    for i, (_, t) in enumerate(ref):
        out.append((torch.cat([g[i][0] for g in gathered if g], dim=0), t))
    return out
