"""Shapes of the remaining benchmark configurations as parity cases, against the CPU oracle on
seeded inputs: colour 3x32x32 training at full width (data-parallel training config) and 3x64x64
with attention at N = 4096 / 1024 / 256 (roofline-stress config); plus a size-independent
property at full width: an eval-mode forward does not depend on how the batch is cut, bit for bit
(every output element is a fixed-order sum over its own sample's data, whatever tile it lands in).

Tolerances as in the golden-vector tests: forward 5e-5 of tensor scale (1e-4 in train mode),
gradients 2e-3 of each tensor's max + 5e-6.
"""
import os

import pytest
import torch

from gpu_util import assert_close
from oracle import ddpm_oracle as O
from seeding import seeded_ints, seeded_state_dict, seeded_tensor
import diffusion_model_amd as pkg

pytestmark = pytest.mark.gpu


def build(variant, cin, base, emb, seed, classes=None):
    U = pkg.make_unet_class(variant)
    m = U(cin, cin, base, emb, num_classes=classes) if variant == 3 and classes else U(cin, cin, base, emb)
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed))
    return m.cuda()


def cpu_sd(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def test_colour_64x64_eval_and_gradients():
    """3x64x64, v3 with labels: attention at N = 4096 (fp32 kernel, 16 channels), 1024, and 256 (split-bf16 kernel)."""
    m = build(3, 3, 16, 32, 11, classes=10)
    sd = cpu_sd(m)
    x = seeded_tensor((2, 3, 64, 64), 1, "x")
    t = torch.tensor([17, 903])
    y = torch.tensor([3, 8])
    m.eval()
    with torch.no_grad():
        out = m(x.cuda(), t.cuda(), y.cuda())
    want, fms, _ = O.unet_forward(sd, 3, x, t, y, train=False)
    assert_close(out.cpu(), want, 5e-5, "64x64 eval eps")
    for i, fm in enumerate(m.feature_maps):
        assert_close(fm.cpu(), fms[i], 5e-5, f"64x64 eval fm{i}")
    m.train()
    tgt = seeded_tensor((2, 3, 64, 64), 2, "tgt")
    loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda(), y.cuda()), tgt.cuda())
    loss.backward()
    ref_loss, grads, _, _ = O.loss_and_grads(sd, 3, x, t, tgt, y)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:6]


def test_colour_64x64_full_width_eval():
    """The roofline-stress configuration at its real width: (3,3,64,128,num_classes=10) on 3x64x64. Attention runs at
    N = 4096 with 64 and 128 channels and at N = 1024 with 256 (the split-bf16 kernels), and at N = 256 with 512
    channels; eps-hat and all seven feature maps against the CPU oracle."""
    m = build(3, 3, 64, 128, 13, classes=10).eval()
    sd = cpu_sd(m)
    x = seeded_tensor((2, 3, 64, 64), 3, "x")
    t = torch.tensor([801, 12])
    y = torch.tensor([7, 0])
    with torch.no_grad():
        out = m(x.cuda(), t.cuda(), y.cuda())
        want, fms, _ = O.unet_forward(sd, 3, x, t, y, train=False)
    assert_close(out.cpu(), want, 5e-5, "64x64 full-width eval eps")
    for i, fm in enumerate(m.feature_maps):
        assert_close(fm.cpu(), fms[i], 5e-5, f"64x64 full-width eval fm{i}")


@pytest.mark.parametrize("hw", [32, 28])
def test_base_channels_128_eval_and_gradients(hw):
    """The wide colour model the reference's CIFAR scripts build (UNet(3, 3, 128, 128), second-iteration/main.py:45):
    1024 channels and 128 query channels in the bottleneck, a fused q|k|v projection of 1280 columns."""
    m = build(2, 3, 128, 128, 14)
    sd = cpu_sd(m)
    x, tgt = seeded_tensor((2, 3, hw, hw), 41, "x"), seeded_tensor((2, 3, hw, hw), 141, "tgt")
    t = torch.tensor([650, 3])
    m.eval()
    with torch.no_grad():
        out = m(x.cuda(), t.cuda())
        want, fms, _ = O.unet_forward(sd, 2, x, t, None, train=False)
    assert_close(out.cpu(), want, 5e-5, "base-128 eval eps")
    for i, fm in enumerate(m.feature_maps):
        assert_close(fm.cpu(), fms[i], 5e-5, f"base-128 eval fm{i}")
    m.train()
    loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda()), tgt.cuda())
    loss.backward()
    ref_loss, grads, _, _ = O.loss_and_grads(sd, 2, x, t, tgt)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:6]
    with pytest.raises(NotImplementedError, match="above 128"):
        pkg.make_unet_class(2)(3, 3, 160, 128)


def test_colour_32x32_full_width_train_step():
    """(3,3,64,128) on 3x32x32: loss and every parameter gradient of one eps-MSE step.

    The input seed is one for which the CPU oracle itself is stable: with random full-width weights the
    2x2 max-pool windows hold near-ties, and a tie that flips under a 1e-6 relative perturbation of the
    weights re-routes a gradient and moves the encoder gradients by up to 5 % (seen with other seeds in
    the oracle alone). Under such a perturbation this seed stays within a quarter of the tolerance.
    """
    m = build(2, 3, 64, 128, 12).train()
    sd = cpu_sd(m)
    x, tgt = seeded_tensor((4, 3, 32, 32), 34, "x"), seeded_tensor((4, 3, 32, 32), 134, "tgt")
    t = seeded_ints((4,), 1000, 234, "t")
    loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda()), tgt.cuda())
    loss.backward()
    ref_loss, grads, _, _ = O.loss_and_grads(sd, 2, x, t, tgt)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:6]


_FULL_SIZE_SEEDS = (41, 42, 43, 44)
_full_size_cache = {}


def _full_size_case(seed):
    """Inputs of the configs[1] training step for one seed and the CPU oracle's loss and gradients for them, in float64
    (the referee) and in float32 (what the reference's own arithmetic achieves); computed once per seed and process."""
    if seed not in _full_size_cache:
        m = pkg.make_unet_class(2)(1, 1, 64, 128)
        sd32 = seeded_state_dict(m.state_dict(), seed)
        sd32 = {k: v.clone() for k, v in sd32.items()}
        sd64 = {k: (v.double() if v.is_floating_point() else v) for k, v in sd32.items()}
        gen = torch.Generator(device="cpu").manual_seed(1000 + seed)
        x, tgt = torch.randn((256, 1, 28, 28), generator=gen), torch.randn((256, 1, 28, 28), generator=gen)
        t = torch.randint(0, 1000, (256,), generator=gen)
        # (another test module pins torch to one CPU thread at import, which holds for the whole pytest process: the two
        # oracle runs of a seed take a minute on one thread and ten seconds on the box's CPU share)
        threads = torch.get_num_threads()
        torch.set_num_threads(max(threads, min(16, int(os.environ.get("OMP_NUM_THREADS", "8")))))
        try:
            loss64, g64, _, _ = O.loss_and_grads(sd64, 2, x.double(), t, tgt.double())
            _, g32, _, _ = O.loss_and_grads(sd32, 2, x, t, tgt)
        finally:
            torch.set_num_threads(threads)
        own = {k: (g32[k].double() - g64[k]).abs().max().item() for k in g64}
        _full_size_cache[seed] = (sd32, x, t, tgt, loss64.item(), g64, own)
        print(f"[full-size train step] CPU oracle of seed {seed} done (float64 and float32)", flush=True)
    return _full_size_cache[seed]


@pytest.mark.parametrize("math", ["f16x2", "bf16x3"])
def test_full_size_train_step_is_as_close_to_float64_as_float32_arithmetic_is(math):
    """configs[1] at the bench's size: (1,1,64,128) v2 UNet, 256 x 1x28x28, one eps-MSE step in train mode (batch
    statistics over all 200704 pixels per channel), on FOUR seeds and in both split arithmetics.

    The referee is the CPU oracle in FLOAT64. What the reference's own arithmetic (float32) achieves against it is
    measured too, per tensor, and the HIP path is held to that:
      * the loss within 1e-5 of the float64 one, every seed;
      * per seed, the MEDIAN over the tensors of (HIP error) / (float32 CPU error) at most 1.5: on a typical tensor the
        HIP path is as close to the exact gradient as a float32 CPU implementation (measured 0.7-1.1);
      * every tensor within the usual gradient tolerance (2e-3 of its largest magnitude + 5e-6) or within 40 times the
        float32 CPU result's own distance from float64, whichever is larger. The second clause is for the gradients that
        float32 arithmetic as such cannot pin -- attention gates and query/key projections, sums of 25 million products
        with heavy cancellation: the float32 CPU result is itself up to 0.7 % off on some seeds and exact to 3e-7 on
        others, and the ratio on such a tensor scatters between 3 and 23 over seeds for every arithmetic, the fp32 MFMA
        included -- and for the encoder gradients on seeds where a 2x2 max-pool window holds a near-tie that the float32
        CPU run resolves differently from the float64 one (there the HIP path is usually the closer of the two). The
        tensors whose float32 CPU result is more than 1e-3 of the gradient's scale away from float64 are printed."""
    from diffusion_model_amd._lib import conv_math
    if conv_math() == "fp32":
        pytest.skip("the diagnostic fp32-MFMA build is not held to this (its accumulation error is the largest of the three)")
    report = []
    for seed in _FULL_SIZE_SEEDS:
        sd32, x, t, tgt, loss64, g64, own = _full_size_case(seed)
        m = pkg.make_unet_class(2)(1, 1, 64, 128)
        m.load_state_dict(sd32)
        m = m.cuda().train()
        m._engine.math = math
        loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda()), tgt.cuda())
        loss.backward()
        assert m._engine.math == math                                   # (no range fallback on the way)
        assert abs(loss.item() - loss64) <= 1e-5 * loss64, (seed, loss.item(), loss64)
        ratios, loose, worst = [], [], (0.0, "")
        for k, p in m.named_parameters():
            ref = g64[k]
            scale = ref.abs().max().item()
            err = (p.grad.cpu().double() - ref).abs().max().item()
            usual = 2e-3 * scale + 5e-6
            assert err <= max(usual, 40.0 * own[k]), (math, seed, k, err, usual, own[k], scale)
            if scale >= 1e-9:      # (gradients that are zero in exact arithmetic hold rounding noise on both sides)
                ratio = err / max(own[k], 1e-12 * scale)
                ratios.append(ratio)
                if own[k] > 1e-3 * scale:
                    loose.append(f"{k} ({own[k] / scale:.1e}; HIP {ratio:.1f}x)")
                elif ratio > worst[0]:
                    worst = (ratio, k)
        ratios.sort()
        median = ratios[len(ratios) // 2]
        report.append(f"seed {seed}: median HIP / float32-CPU error {median:.2f}, worst {worst[0]:.1f} ({worst[1]}); "
                      f"float32 CPU itself more than 1e-3 off on {len(loose)} tensors: {', '.join(loose[:4])}{' ...' if len(loose) > 4 else ''}")
        assert median <= 1.5, (math, seed, median)
        del m
    print(f"\n[full-size train step, {math}] " + "\n    ".join(report))


@pytest.mark.parametrize("variant,hw", [(1, 28), (2, 28), (3, 32)])
def test_eval_forward_is_independent_of_the_batch_cut(variant, hw):
    m = build(variant, 1, 64, 128, 20 + variant, classes=10 if variant == 3 else None).eval()
    x = seeded_tensor((24, 1, hw, hw), 6, "x").cuda()
    t = seeded_ints((24,), 1000, 7, "t").cuda()
    y = seeded_ints((24,), 10, 8, "y").cuda() if variant == 3 else None
    with torch.no_grad():
        def run(sl):
            return m(x[sl], t[sl], y[sl]) if variant == 3 else m(x[sl], t[sl])
        whole = run(slice(0, 24))
        parts = torch.cat([run(slice(0, 5)), run(slice(5, 16)), run(slice(16, 24))])
    assert torch.equal(whole, parts)


@pytest.mark.parametrize("cin,hw,batch", [(1, 32, 512), (3, 64, 1024)])
def test_full_size_eval_matches_small_batches(cin, hw, batch):
    """The two sampling configurations at the per-GPU batch the bench runs them at (512 x 1x32x32; 1024 x 3x64x64, whose
    128-channel tensors at full resolution exceed 2 GiB): samples from the beginning, the middle and the very end of the
    batch come out bit for bit as when they are run alone in a small batch. What a wrong 32-bit offset, a tile that
    runs past its tensor or a kernel choice that depends on the row count would break."""
    m = build(3, cin, 64, 128, 31, classes=10).eval()
    gen = torch.Generator(device="cpu").manual_seed(5)
    x = torch.randn((batch, cin, hw, hw), generator=gen).cuda()
    t = torch.randint(0, 1000, (batch,), generator=gen).cuda()
    y = torch.randint(0, 10, (batch,), generator=gen).cuda()
    with torch.no_grad():
        whole = m(x, t, y)
        assert torch.isfinite(whole).all()
        for sl in (slice(0, 6), slice(batch // 2 - 3, batch // 2 + 4), slice(batch - 9, batch)):
            part = m(x[sl].contiguous(), t[sl].contiguous(), y[sl].contiguous())
            assert torch.equal(whole[sl], part), f"samples {sl} differ from their small-batch run"


def test_full_size_sampling_trajectory_matches_small_batches():
    """configs[2] at its per-GPU batch (512 x 1x32x32, labels): a T = 10 ancestral trajectory with injected noise, all ten
    stages; the first, middle and last samples equal, bit for bit, the same trajectory run on those samples alone
    (the fused final-conv + reverse-step kernel, the stage copies and the label path at full size)."""
    B, T = 512, 10
    m = build(3, 1, 64, 128, 32, classes=10).eval()
    betas = pkg.BetaScheduler(T, device="cuda").get_schedule("linear")
    dm = pkg.make_diffusion_class(3)(m, betas, T)
    gen = torch.Generator(device="cpu").manual_seed(9)
    shape = (B, 1, 32, 32)
    noise = (torch.randn(shape, generator=gen), [torch.randn(shape, generator=gen) for _ in range(T - 1)])
    labels = torch.randint(0, 10, (B,), generator=gen).cuda()
    with torch.no_grad():
        whole = dm.sample(shape, "cuda", labels=labels, noise=noise)
        assert len(whole) == 10 and all(torch.isfinite(x).all() for x, _ in whole)
        for sl in (slice(0, 5), slice(253, 259), slice(B - 7, B)):
            part = dm.sample((sl.stop - sl.start,) + shape[1:], "cuda", labels=labels[sl].contiguous(),
                             noise=(noise[0][sl], [z[sl] for z in noise[1]]))
            for (a, ta), (b, tb) in zip(whole, part):
                assert ta == tb and torch.equal(a[sl], b), f"stage t={ta}, samples {sl}"


def test_two_stream_backward_matches_one_stream():
    """The weight gradients run on a second stream beside the data-gradient chain: same gradients as the
    single-stream order (weight gradients are split-K atomic sums, so equal to rounding, not bit for bit),
    step after step (the second step reuses memory the first one released)."""
    from diffusion_model_amd.engine_train import TrainMixin
    x = seeded_tensor((8, 1, 28, 28), 5, "x")
    eps = seeded_tensor((8, 1, 28, 28), 6, "eps")
    t = torch.tensor([3, 999, 500, 17, 0, 250, 731, 64])
    results = []
    saved = TrainMixin._use_side
    try:
        for side in (False, True):
            TrainMixin._use_side = side
            m = build(2, 1, 32, 64, 21)
            m.train()
            per_step = []
            for _ in range(3):
                m.zero_grad()
                loss = torch.nn.functional.mse_loss(m(x.cuda(), t.cuda()), eps.cuda())
                loss.backward()
                torch.cuda.synchronize()
                per_step.append({k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
            results.append(per_step)
    finally:
        TrainMixin._use_side = saved
    for a, b in zip(*results):
        for k in a:
            assert_close(b[k], a[k], 2e-5, f"two-stream gradient {k}")
