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


def _generator_digests(device):
    """(state of torch's CPU generator, state of the generator of `device` if it is a GPU), as hashable digests."""
    import hashlib
    cpu = hashlib.sha1(torch.get_rng_state().numpy().tobytes()).hexdigest()
    dev = torch.device(device)
    gpu = None
    if dev.type == "cuda":
        gpu = hashlib.sha1(torch.cuda.get_rng_state(dev).numpy().tobytes()).hexdigest()
    return cpu, gpu


def decorrelate_rng(device, group=None):
    """Make sure no two ranks draw the same noise. Scripts usually seed every process alike (torch.manual_seed(0)), and
    ranks that then draw their own x_T and z would all produce the SAME samples. sample() draws x_T -- and, with
    in_kernel_noise, the seed of the per-step noise -- from torch's CPU generator and the per-step noise from the
    generator of `device`, so BOTH are compared: the ranks exchange digests of the generators' current states, and
    every rank whose state another rank shares reseeds that generator with (its seed + rank); rank 0 keeps the sequence
    a single process would have had. Called before every sharded sampling call that draws its own noise: a script
    that reseeds all ranks alike between two calls is caught again. Returns the names of the generators reseeded."""
    world, rank = _world(group)
    if world == 1:
        return ()
    dev = torch.device(device)
    mine = _generator_digests(dev)
    everyone = [None] * world
    dist.all_gather_object(everyone, mine, group=group)
    reseeded = []
    if len({d[0] for d in everyone}) < world:
        if rank:
            torch.default_generator.manual_seed((int(torch.initial_seed()) + rank) % (1 << 63))
        reseeded.append("cpu")
    if dev.type == "cuda" and len({d[1] for d in everyone}) < world:
        if rank:
            with torch.cuda.device(dev):
                torch.cuda.manual_seed((int(torch.cuda.initial_seed()) + rank) % (1 << 63))
        reseeded.append("cuda")
    return tuple(reseeded)


def sample_sharded(diffusion, x_shape, device, gather=True, noise=None, group=None, **sample_kwargs):
    """Run diffusion.sample on this rank's slice of the batch axis.

    noise: optional (x_T, [z ...]) for the WHOLE batch; each rank injects its slice, so that the
    sharded result equals the single-process one sample for sample. Without it every rank draws its
    own noise, from generators that decorrelate_rng has made distinct.
    Returns the reference's stage list; with gather=True every rank gets the whole batch
    (concatenated in rank order), otherwise only its own slice.
    """
    world, rank = _world(group)
    lo, hi = shard_range(x_shape[0], world, rank)
    local_shape = (hi - lo,) + tuple(x_shape[1:])
    if noise is not None:
        sample_kwargs["noise"] = (noise[0][lo:hi], [z[lo:hi] for z in noise[1]])
    else:
        decorrelate_rng(device, group)
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


def world_size(group=None):
    return _world(group)[0]


class BucketedGradientAverager:
    """Data-parallel gradient averaging that overlaps with the backward pass.

    The UNet's hand-written backward (engine_train.TrainMixin.backward) asks this object for the storage of every
    parameter gradient, in the order it produces them (decoder first). The storage is consecutive slices of ONE
    persistent flat buffer; at the backward's flush points every slice handed out so far has all its kernels enqueued,
    and once `bucket_bytes` of such gradients have accumulated they are all-reduced asynchronously (RCCL on its own
    stream) while the backward goes on. finish() waits for the buckets, scales to the mean and points every
    parameter's .grad at its slice: no concatenation, no copy back, and only the last bucket's all-reduce is exposed.

    The first step runs in recording mode (the gradients are ordinary tensors, averaged by average_gradients) to
    learn the sequence of slice sizes; the flat buffer is built from it. Usage per step:
        loss.backward(); averager.finish(); optimizer.step()
    """

    ALIGN = 64      # floats: every slice starts on a 256-byte boundary

    def __init__(self, model, bucket_bytes=8 << 20, group=None):
        self.model, self.group = model, group
        self.bucket_floats = max(1, int(bucket_bytes) // 4)
        self.engine = model._engine
        self.sizes = None          # slice sizes (floats, aligned) in the order of the backward, once recorded
        self.flat = None
        self.post_scale = None     # one-element device tensor: factor the backward still owes the slices (power-of-two gradient scaling)
        self._recording = []
        self._reset()

    def owns(self, t):
        """True when `t` is a slice of the flat buffer (False in the recording step, whose gradients are plain tensors)."""
        return self.flat is not None and t.untyped_storage().data_ptr() == self.flat.untyped_storage().data_ptr()

    def attach(self):
        self.engine.grad_sink = self
        self.engine.dist_group = self.group     # global-batch BatchNorm and the gradient scale reduce over the same ranks
        return self

    def detach(self):
        if self.engine.grad_sink is self:
            self.engine.grad_sink = None

    def _reset(self):
        self._i, self._off, self._sent, self._work, self._grads = 0, 0, 0, [], None

    # ---- called by the engine during the backward ------------------------------------------------
    def alloc(self, numel, dev):
        padded = -(-int(numel) // self.ALIGN) * self.ALIGN
        if self.flat is None:                                  # recording mode
            self._recording.append(padded)
            return torch.empty(numel, device=dev, dtype=torch.float32)
        if self._i >= len(self.sizes) or self.sizes[self._i] != padded:
            raise RuntimeError("the backward pass asked for gradient storage in a different order or size than when the "
                               "averager recorded it (model or input structure changed): build a new BucketedGradientAverager")
        t = self.flat[self._off:self._off + numel]
        self._i += 1
        self._off += padded
        return t

    def flush(self, engine, force=False):
        if self.flat is None or self._off == self._sent:
            return
        if not force and self._off - self._sent < self.bucket_floats:
            return
        bucket = self.flat[self._sent:self._off]
        self._sent = self._off
        if not (dist.is_available() and dist.is_initialized()):
            return
        if bucket.is_cuda and _host_staged(self.group):       # gloo rehearsal: synchronous, through the host
            side = getattr(engine, "_s2", None)
            if side is not None:
                torch.cuda.current_stream().wait_stream(side)  # weight gradients are written on the side stream
            all_reduce_(bucket, group=self.group)
            self._work.append(None)
            return
        if bucket.is_cuda:
            # gradients are written by two streams (data-gradient chain on the current one, weight gradients on the
            # engine's side stream): the collective is issued from the side stream after it has caught up with the
            # current one, so the data-gradient chain itself never waits for it
            main = torch.cuda.current_stream()
            side = getattr(engine, "_s2", None)
            if side is not None and side != main:
                side.wait_stream(main)
                with torch.cuda.stream(side):
                    self._work.append(dist.all_reduce(bucket, group=self.group, async_op=True))
                return
        self._work.append(dist.all_reduce(bucket, group=self.group, async_op=True))

    def takes_over(self, grads):
        """End of the backward: True when the gradients live in the flat buffer (finish() will assign .grad)."""
        if self.flat is None:
            return False
        self._grads = grads
        return True

    # ---- called by the training loop after loss.backward() ---------------------------------------
    def finish(self):
        world, _ = _world(self.group)
        if self.flat is None:
            # recording step: plain tensors (already unscaled by the backward), one flat all-reduce; then build the persistent buffer
            self.post_scale = None
            average_gradients(list(self.model.parameters()), group=self.group)
            if self._recording:
                self.sizes = list(self._recording)
                dev = next(self.model.parameters()).device
                self.flat = torch.zeros(sum(self.sizes), device=dev, dtype=torch.float32)
            self._recording = []
            self._reset()
            return
        if self._grads is None:
            raise RuntimeError("finish() without a backward pass through the UNet")
        self.flush(self.engine, force=True)
        for w in self._work:
            if w is not None:
                w.wait()                                       # the current stream waits for the collective; the host does not
        if world > 1:
            self.flat.div_(world)
        if self.post_scale is not None:
            self.flat.mul_(self.post_scale)                    # exact: a power of two
            self.post_scale = None
        for name, p in self.model.named_parameters():
            g = self._grads.get(name)
            if g is not None:
                p.grad = g
        self._reset()
