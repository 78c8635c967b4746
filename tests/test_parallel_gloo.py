"""world_size-2 checks of the multi-GPU helpers on CPU (gloo): batch sharding, gradient averaging,
parameter broadcast, sharded sampling == single-process sampling."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from diffusion_model_amd import parallel
from trace_util import load_traces


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 4096, 4099):
        for world in (1, 2, 3, 8):
            spans = [parallel.shard_range(total, world, r) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1


class ToyDiffusion:
    """Deterministic stand-in with the reference's sample() contract (list of (cpu tensor, t))."""

    def sample(self, x_shape, device="cpu", noise=None):
        if noise is None:                       # own noise from torch's default generator, as the reference's sample() draws it
            noise = (torch.randn(x_shape), [torch.randn(x_shape) for _ in range(3)])
        x = noise[0].clone()
        stages = []
        for i, z in enumerate(noise[1]):
            x = 0.9 * x + 0.1 * z + 0.01 * x.mean(dim=(1, 2, 3), keepdim=True)      # per-sample only
            stages.append((x.clone(), len(noise[1]) - 1 - i))
        return stages


class ToyEngine:
    """Stands in for the UNet's backward schedule: asks the averager for gradient storage in a fixed order, fills it,
    and marks flush points, exactly as engine_train.TrainMixin.backward does (CPU tensors, no streams)."""
    grad_sink = None
    _s2 = None

    def __init__(self, model):
        self.m = model

    def backward(self, scale):
        # as the real backward under the fp16-pair arithmetic: the incoming gradient's largest magnitude, agreed on by all
        # ranks (MAX), gives a power-of-two factor s; the gradients are produced times s (and summed across ranks in that
        # state), and whoever owns their storage multiplies by 1 / s once at the end
        amax = torch.tensor([scale * 3e-6])
        if dist.is_available() and dist.is_initialized():
            parallel.all_reduce_(amax, op=dist.ReduceOp.MAX)
        s = 2.0 ** -int(torch.floor(torch.log2(amax)).item())
        grads, own = {}, []
        for i, (name, p) in enumerate(reversed(list(self.m.named_parameters()))):
            g = self.grad_sink.alloc(p.numel(), p.device).view(p.shape) if self.grad_sink else torch.empty_like(p)
            if self.grad_sink is None or not self.grad_sink.owns(g):
                own.append(g)
            g.copy_(torch.full_like(p, scale * (i + 1) * s))
            grads[name] = g
            if self.grad_sink:
                self.grad_sink.flush(self)
        for g in own:
            g.mul_(1.0 / s)
        if self.grad_sink is not None:
            self.grad_sink.post_scale = torch.tensor([1.0 / s])
        if self.grad_sink is not None and self.grad_sink.takes_over(grads):
            return
        for name, p in self.m.named_parameters():
            p.grad = grads[name]


class ToyModel(torch.nn.Module):
    def __init__(self):
        super().__init__()
        self.a, self.b, self.c = torch.nn.Linear(7, 33), torch.nn.Linear(33, 129), torch.nn.Linear(129, 5)
        self._engine = ToyEngine(self)


def _check_bucketed_averager(rank, world):
    m = ToyModel()
    avg = parallel.BucketedGradientAverager(m, bucket_bytes=2048).attach()
    names = [n for n, _ in m.named_parameters()]
    mean_scale = sum(r + 1 for r in range(world)) / world
    for step in range(3):                                   # step 0 records the layout, steps 1-2 use the flat buffer
        m.zero_grad()
        m._engine.backward(float(rank + 1))
        avg.finish()
        for i, name in enumerate(reversed(names)):
            g = dict(m.named_parameters())[name].grad
            assert torch.allclose(g, torch.full_like(g, mean_scale * (i + 1))), (step, name)
        if step > 0:
            assert all(p.grad.untyped_storage().data_ptr() == avg.flat.untyped_storage().data_ptr() for p in m.parameters())
            assert all(p.grad.data_ptr() % 256 == avg.flat.data_ptr() % 256 for p in m.parameters())   # 256-byte aligned slices
    # a backward that asks for other sizes than the recorded ones is refused, not silently mis-sliced
    m.c = torch.nn.Linear(129, 6)
    with pytest.raises(RuntimeError, match="different order or size"):
        m._engine.backward(1.0)
    avg.detach()
    assert m._engine.grad_sink is None


class ReplayEngine:
    """What BucketedGradientAverager needs of the UNet's engine, on CPU tensors: no side stream."""
    grad_sink = None
    dist_group = None
    _s2 = None


class ReplayModel(torch.nn.Module):
    """One flat parameter per gradient-storage request of the recorded backward, in the order of the requests."""

    def __init__(self, sizes):
        super().__init__()
        self.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(n)) for n in sizes])
        self._engine = ReplayEngine()


def _check_real_backward_trace(rank, world):
    """The REAL sequence of storage requests and flush points of engine_train's backward (recorded on the GPU from the
    full-width second-iteration UNet: 12.3 M parameters in 100-odd slices, tests/golden/backward_trace.json), replayed
    through the real averager with 8 MiB buckets: every gradient comes out as the mean over the ranks, the buckets are
    cut only at flush points, all but the last hold at least 8 MiB, and together they cover the flat buffer once."""
    trace = load_traces()["v2_full_width_28"]
    events = [tuple(e) for e in trace["events"]]
    sizes = [e[1] for e in events if e[0] == "alloc"]
    assert sum(sizes) >= trace["parameters"] and len(sizes) > 90
    m = ReplayModel(sizes)
    avg = parallel.BucketedGradientAverager(m, bucket_bytes=8 << 20).attach()
    sent = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        if avg.flat is not None and t.untyped_storage().data_ptr() == avg.flat.untyped_storage().data_ptr():
            sent.append((t.storage_offset(), t.numel()))
        return real_all_reduce(t, *a, **k)

    names = [n for n, _ in m.named_parameters()]
    mean = sum(r + 1 for r in range(world)) / world
    dist.all_reduce = counting_all_reduce
    try:
        for step in range(3):                       # step 0 records the layout, steps 1-2 run in buckets
            del sent[:]
            grads, i, flush_offsets = {}, 0, []
            for ev in events:
                if ev[0] == "alloc":
                    g = avg.alloc(ev[1], torch.device("cpu"))
                    g.fill_(float((rank + 1) * (i % 7 + 1)))
                    grads[names[i]] = g
                    i += 1
                else:
                    avg.flush(m._engine)
                    flush_offsets.append(avg._off)
            if not avg.takes_over(grads):
                for n, p in m.named_parameters():
                    p.grad = grads[n]
            avg.finish()
            for i, (n, p) in enumerate(m.named_parameters()):
                assert torch.all(p.grad == mean * (i % 7 + 1)), (step, n)
            if step > 0:
                assert len(sent) >= 5, sent                                            # 49 MB in 8 MiB buckets
                assert sent[0][0] == 0 and all(a[0] + a[1] == b[0] for a, b in zip(sent, sent[1:]))
                assert sent[-1][0] + sent[-1][1] == avg.flat.numel() == sum(avg.sizes)
                assert all(n >= (8 << 20) // 4 for _, n in sent[:-1])
                assert all(off + n in flush_offsets or (off + n == avg.flat.numel()) for off, n in sent)   # cut at flush points only
    finally:
        dist.all_reduce = real_all_reduce
    avg.detach()


def _worker(rank, world, port, tmp):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        # gradient averaging
        torch.manual_seed(0)
        lin = torch.nn.Linear(5, 3)
        parallel.broadcast_parameters(lin)
        w0 = lin.weight.detach().clone()
        x = torch.full((4, 5), float(rank + 1))
        lin(x).sum().backward()
        local = [p.grad.clone() for p in lin.parameters()]
        parallel.average_gradients(list(lin.parameters()))
        mean_scale = sum(r + 1 for r in range(world)) / world
        assert torch.allclose(lin.weight.grad, local[0] / (rank + 1) * mean_scale)
        assert torch.allclose(lin.bias.grad, local[1])                 # identical on all ranks
        assert torch.equal(lin.weight.detach(), w0)
        _check_bucketed_averager(rank, world)
        _check_real_backward_trace(rank, world)
        # sharded sampling
        g = torch.Generator().manual_seed(1)
        shape = (5, 1, 4, 4)                                           # odd batch: uneven shards
        noise = (torch.randn(shape, generator=g), [torch.randn(shape, generator=g) for _ in range(3)])
        whole = ToyDiffusion().sample(shape, noise=noise)
        got = parallel.sample_sharded(ToyDiffusion(), shape, "cpu", noise=noise)
        assert [t for _, t in got] == [t for _, t in whole]
        for (a, _), (b, _) in zip(got, whole):
            assert torch.equal(a, b)
        mine = parallel.sample_sharded(ToyDiffusion(), shape, "cpu", gather=False, noise=noise)
        lo, hi = parallel.shard_range(5, world, rank)
        assert torch.equal(mine[-1][0], whole[-1][0][lo:hi])
        # own noise: every rank was seeded alike above (torch.manual_seed(0)), yet the shards must not repeat each other
        torch.manual_seed(0)
        own = parallel.sample_sharded(ToyDiffusion(), (4, 1, 4, 4), "cpu")
        final = own[-1][0]
        assert final.shape[0] == 4 and not torch.equal(final[:2], final[2:])
        torch.manual_seed(0)
        expect_rank0 = ToyDiffusion().sample((2, 1, 4, 4))[-1][0]
        assert torch.equal(final[:2], expect_rank0)                    # rank 0 keeps the single-process sequence
        open(os.path.join(tmp, f"ok{rank}"), "w").close()
    finally:
        dist.destroy_process_group()


def test_world_size_two_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert sorted(os.listdir(tmp_path)) == ["ok0", "ok1"]
