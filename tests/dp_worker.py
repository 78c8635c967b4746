"""Worker of tests/test_gpu_dp.py: one rank of a 2-rank data-parallel job whose ranks SHARE GPU 0 (gloo moves the
collectives through the host; the GPU work is the real HIP path). Checks, against the reference's golden vectors:
  1. global-batch BatchNorm + averaged gradients of a sharded batch == the reference's full-batch gradients;
  2. the bucketed averager (recording step, then buckets from inside the backward) == the flat all-reduce;
  3. 10 Adam steps sharded 2 + 2 reproduce the reference's single-process loss trace (batch 4);
  4. sharded sampling with sliced injected noise == unsharded sampling, bit for bit;
  5. ranks that were seeded alike still draw different noise on the GPU (torch's generators and the in-kernel stream);
  6. global-batch BatchNorm refuses shards of different sizes;
  7. a full-width network cut 2 + 2 with global-batch BatchNorm == one process on the whole batch (loss, gradients, running statistics).
Prints DP_WORKER_OK on rank 0."""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

from seeding import seeded_ints, seeded_state_dict, seeded_tensor  # noqa: E402
import diffusion_model_amd as pkg  # noqa: E402
from diffusion_model_amd import parallel  # noqa: E402
from diffusion_model_amd.optim import Adam  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")


def build(variant, seed):
    U = pkg.make_unet_class(variant)
    m = U(1, 1, 8, 16) if variant < 3 else U(3, 3, 8, 16, num_classes=10)
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed))
    return m.cuda()


def mean_over_ranks(v):
    t = torch.tensor([v], dtype=torch.float64)
    dist.all_reduce(t)
    return t.item() / dist.get_world_size()


def check_grads(m, g, tag, what):
    worst = []
    for k, p in m.named_parameters():
        ref = torch.from_numpy(g[f"{tag}_grad::{k}"])
        got = p.grad.cpu() if p.grad is not None else torch.zeros_like(ref)
        err = (got - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, (what, worst[:6])


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    assert world == 2
    torch.cuda.set_device(0)

    # ---- 1 + 2: gradients of a sharded batch with global-batch BatchNorm ---------------------------------------
    g = np.load(os.path.join(GOLDEN, "tiny_unet_v2.npz"))
    x, t, tgt = (torch.from_numpy(g[f"hw28_{k}"]) for k in ("x", "t", "tgt"))
    lo, hi = parallel.shard_range(x.shape[0], world, rank)
    for mode in ("flat", "bucketed"):
        m = build(2, 2).train()
        m._engine.sync_bn = True
        parallel.broadcast_parameters(m)
        avg = parallel.BucketedGradientAverager(m, bucket_bytes=4096).attach() if mode == "bucketed" else None
        for step in range(3 if avg else 1):          # bucketed: step 0 records the layout, steps 1-2 run in buckets
            m.zero_grad()
            loss = torch.nn.MSELoss()(m(x[lo:hi].cuda(), t[lo:hi].cuda()), tgt[lo:hi].cuda())
            loss.backward()
            if avg:
                avg.finish()
                assert (avg.flat is not None) and (step == 0 or all(p.grad.untyped_storage().data_ptr() == avg.flat.untyped_storage().data_ptr()
                                                                       for p in m.parameters()))
            else:
                parallel.average_gradients(list(m.parameters()))
            global_loss = mean_over_ranks(loss.item())
            assert abs(global_loss - g["hw28_train_loss"][0]) <= 1e-4 * g["hw28_train_loss"][0], (mode, step, global_loss)
            check_grads(m, g, "hw28", f"{mode} step {step}")
            if step == 0:                              # running statistics after ONE update are the full batch's
                for k, v in m.state_dict().items():
                    if "running_" in k:
                        ref = torch.from_numpy(g[f"hw28_buf::{k}"])
                        assert (v.cpu() - ref).abs().max().item() <= 1e-5 * max(1.0, ref.abs().max().item()), k

    # per-rank statistics (the default) do NOT reproduce the full-batch gradients: the switch matters
    m = build(2, 2).train()
    loss = torch.nn.MSELoss()(m(x[lo:hi].cuda(), t[lo:hi].cuda()), tgt[lo:hi].cuda())
    loss.backward()
    parallel.average_gradients(list(m.parameters()))
    ref = torch.from_numpy(g["hw28_grad::enc1.double_conv.0.weight"])
    assert (m.enc1.double_conv[0].weight.grad.cpu() - ref).abs().max().item() > 2e-3 * ref.abs().max().item()

    # ---- 3: ten Adam steps, batch 4 cut 2 + 2 --------------------------------------------------------------------
    gt = np.load(os.path.join(GOLDEN, "training_v2.npz"))
    m = build(2, 2).train()
    m._engine.sync_bn = True
    avg = parallel.BucketedGradientAverager(m, bucket_bytes=16384).attach()
    dm = pkg.make_diffusion_class(2)(m, pkg.BetaScheduler(1000, device="cuda").get_schedule("linear"), 1000)
    opt = Adam(m.parameters(), lr=2e-4)
    losses = []
    for s in range(10):
        x0 = seeded_tensor((4, 1, 28, 28), 300, f"x0_{s}", "uniform")[2 * rank:2 * rank + 2].cuda()
        noise = seeded_tensor((4, 1, 28, 28), 300, f"n_{s}")[2 * rank:2 * rank + 2].cuda()
        ts = seeded_ints((4,), 1000, 300, f"t_{s}")[2 * rank:2 * rank + 2].cuda()
        pred = m(dm.add_noise(x0, noise, ts), ts)
        loss = torch.nn.MSELoss()(pred, noise)
        opt.zero_grad()
        loss.backward()
        avg.finish()
        opt.step()
        losses.append(mean_over_ranks(loss.item()))
    np.testing.assert_allclose(losses, gt["losses"], rtol=2e-3)
    sd = m.state_dict()
    for k in ("final_conv.weight", "time_fc3.weight", "dec2.up.weight", "enc2.attention.gamma", "bottleneck.0.double_conv.1.weight"):
        ref = torch.from_numpy(gt[f"final::{k}"])
        assert (sd[k].cpu() - ref).abs().max().item() <= 2e-3 * max(1.0, ref.abs().max().item()), k
    # every rank holds the same parameters after the same averaged updates
    probe = torch.cat([p.detach().reshape(-1)[:64].cpu() for p in m.parameters()])
    other = probe.clone()
    dist.broadcast(other, src=0)
    assert torch.equal(probe, other)

    # ---- 4: sharded sampling == unsharded sampling ---------------------------------------------------------------
    m3 = build(3, 3).eval()
    T, shape = 10, (4, 3, 28, 28)
    dm3 = pkg.make_diffusion_class(3)(m3, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
    xT = seeded_tensor(shape, 9, "xT")
    zs = [seeded_tensor(shape, 9, f"z{i}") for i in range(T - 1)]
    labels = torch.tensor([5])
    with torch.no_grad():
        sharded = parallel.sample_sharded(dm3, shape, "cuda", noise=(xT, zs), labels=labels, cfg_scale=1.5)
        whole = dm3.sample(shape, "cuda", noise=(xT, zs), labels=labels, cfg_scale=1.5)
    assert [s[1] for s in sharded] == [s[1] for s in whole]
    assert all(torch.equal(a[0], b[0]) for a, b in zip(sharded, whole))

    # ---- 5: ranks seeded alike must not produce the same samples -------------------------------------------------
    # (x_T and the in-kernel noise seed come from torch's CPU generator, the per-step noise from the GPU's)
    for in_kernel in (False, True):
        for round_ in range(2):                      # the second round reseeds all ranks alike AGAIN: caught again
            torch.manual_seed(0)                     # seeds the CPU and the GPU generator, the same on both ranks
            with torch.no_grad():
                both = parallel.sample_sharded(dm3, shape, "cuda", labels=labels, in_kernel_noise=in_kernel)
            first, last = both[0][0], both[-1][0]    # whole batch, rank 0's samples first
            assert first.shape[0] == shape[0]
            assert not torch.equal(first[:2], first[2:]), ("same x_T / first-stage samples on both ranks", in_kernel, round_)
            assert not torch.equal(last[:2], last[2:]), ("same final samples on both ranks", in_kernel, round_)
    # rank 0 keeps the sequence a single process seeded the same way would have had
    torch.manual_seed(0)
    with torch.no_grad():
        both = parallel.sample_sharded(dm3, shape, "cuda", labels=labels)
        torch.manual_seed(0)
        alone = dm3.sample((2,) + shape[1:], "cuda", labels=labels)
    if rank == 0:
        assert torch.equal(both[-1][0][:2], alone[-1][0])

    # ---- 6: global-batch BatchNorm refuses shards of different sizes ---------------------------------------------
    m = build(2, 2).train()
    m._engine.sync_bn = True
    n_mine = 2 if rank == 0 else 1
    try:
        m(x[:n_mine].cuda(), t[:n_mine].cuda())
        raise AssertionError("uneven shards were accepted")
    except ValueError as e:
        assert "same number of samples" in str(e), e

    # ---- 7: full-width network (the halo kernel's epilogue statistics feed the all-reduced sums) -------------------
    # two ranks with global-batch BatchNorm against ONE process on the whole batch, both on the HIP path
    U = pkg.make_unet_class(2)
    wide = U(1, 1, 64, 128)
    wide.load_state_dict(seeded_state_dict(wide.state_dict(), 8))
    wide = wide.cuda().train()
    xb, tb = seeded_tensor((4, 1, 28, 28), 70, "x"), seeded_ints((4,), 1000, 70, "t")
    tg = seeded_tensor((4, 1, 28, 28), 71, "tgt")
    sd0 = {k: v.clone() for k, v in wide.state_dict().items()}
    whole_loss = torch.nn.MSELoss()(wide(xb.cuda(), tb.cuda()), tg.cuda())
    whole_loss.backward()
    whole = {k: p.grad.clone() for k, p in wide.named_parameters()}
    whole_stats = {k: v.clone() for k, v in wide.state_dict().items() if "running_" in k}
    wide.load_state_dict(sd0)
    wide.zero_grad()
    wide._engine.sync_bn = True
    lo, hi = parallel.shard_range(4, world, rank)
    loss = torch.nn.MSELoss()(wide(xb[lo:hi].cuda(), tb[lo:hi].cuda()), tg[lo:hi].cuda())
    loss.backward()
    parallel.average_gradients(list(wide.parameters()))
    assert abs(mean_over_ranks(loss.item()) - whole_loss.item()) <= 1e-5 * abs(whole_loss.item())
    # (+ 1e-6: the biases in front of a BatchNorm have gradients that are zero up to rounding noise)
    worst = sorted(((p.grad - whole[k]).abs().max().item() / (2e-4 * whole[k].abs().max().item() + 1e-6), k)
                   for k, p in wide.named_parameters())[::-1]
    assert worst[0][0] <= 1.0, ("full-width sync-BN against one process", worst[:5])
    for k, v in wide.state_dict().items():
        if "running_" in k:
            assert (v - whole_stats[k]).abs().max().item() <= 1e-6 * max(1.0, whole_stats[k].abs().max().item()), k

    dist.barrier()
    if rank == 0:
        print("DP_WORKER_OK", flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
        with open(os.path.join(ROOT, "gpurun_out", f"dp_worker_rank{os.environ.get('RANK', '0')}.err"), "w") as f:
            traceback.print_exc(file=f)
        raise
