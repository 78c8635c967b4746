"""The sampling loop's fused last kernel (final 1x1 conv + reverse step, optional in-kernel noise) and the hand-off
kernels that follow the loop (to [0, 1] images, image grid).

Bars: the fused kernel is BIT-IDENTICAL to the two-kernel path (ddpm_conv1x1_final_fwd then ddpm_reverse_step) on the
same operands, whole sampling trajectories included; in-kernel noise equals the same Philox stream drawn by
ddpm_philox_normal and injected; to_images equals the torch expression bit for bit; make_grid equals the oracle's
restatement of torchvision's algorithm (torchvision is not installed: parity unpinned for that one function)."""
import ctypes as C
import math

import numpy as np
import pytest
import torch

from gpu_util import P, call, rnd, stream, to_nhwc
from oracle import ddpm_oracle as O
from seeding import seeded_state_dict, seeded_tensor
import diffusion_model_amd as pkg
from diffusion_model_amd import process

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("B,H,W,c,cout", [(3, 11, 13, 64, 1), (2, 32, 32, 64, 3), (1, 7, 5, 8, 1), (5, 28, 28, 72, 3)])
@pytest.mark.parametrize("mode", ["none", "z", "philox", "cfg_low", "cfg_high"])
def test_fused_kernel_equals_final_conv_then_reverse_step(B, H, W, c, cout, mode):
    feat = to_nhwc(rnd((B, c, H, W), 1))
    w, b = rnd((cout, c), 2, c ** -0.5).cuda(), rnd((cout,), 3).cuda()
    x_t, z, eps_c = rnd((B, cout, H, W), 4).cuda(), rnd((B, cout, H, W), 5).cuda(), rnd((B, cout, H, W), 6).cuda()
    c1, c2, c3 = 1.0101, 0.0123, 0.1414
    n = x_t.numel()
    seed, callno = 0x1234567890ABCDEF, 977
    cfg_w = {"cfg_low": 0.3, "cfg_high": 1.5}.get(mode, 0.0)
    # two-kernel path
    eps = torch.empty_like(x_t)
    call("ddpm_conv1x1_final_fwd", P(feat), c, P(w), P(b), P(eps), B, H, W, c, cout, stream())
    zz = None
    if mode == "philox":
        zz = torch.empty_like(x_t)
        call("ddpm_philox_normal", P(zz), n, C.c_uint64(seed), C.c_uint64(callno), stream())
    elif mode != "none":
        zz = z
    want = torch.empty_like(x_t)
    if mode.startswith("cfg"):      # this pass is the unconditional prediction; eps_c the conditional one
        call("ddpm_reverse_step", P(x_t), P(eps_c), P(eps), cfg_w, P(zz), c1, c2, c3, P(want), n, stream())
    else:
        call("ddpm_reverse_step", P(x_t), P(eps), None, 0.0, P(zz), c1, c2, c3, P(want), n, stream())
    # fused
    got, eps_out = torch.empty_like(x_t), torch.empty_like(x_t)
    nm = {"none": 0, "philox": 2}.get(mode, 1)
    call("ddpm_final_conv_reverse_step", P(feat), c, P(w), P(b), P(x_t), P(eps_c) if mode.startswith("cfg") else None, cfg_w, nm,
         P(z) if nm == 1 else None, C.c_uint64(seed), C.c_uint64(callno), c1, c2, c3, P(got), P(eps_out), B, H, W, c, cout, stream())
    assert torch.equal(got, want)
    assert torch.equal(eps_out, eps)
    # in place
    x_alias = x_t.clone()
    call("ddpm_final_conv_reverse_step", P(feat), c, P(w), P(b), P(x_alias), P(eps_c) if mode.startswith("cfg") else None, cfg_w, nm,
         P(z) if nm == 1 else None, C.c_uint64(seed), C.c_uint64(callno), c1, c2, c3, P(x_alias), None, B, H, W, c, cout, stream())
    assert torch.equal(x_alias, want)


def test_philox_stream_is_standard_normal_and_addressable():
    n = 1 << 22
    a, b, c = (torch.empty(n, device="cuda") for _ in range(3))
    call("ddpm_philox_normal", P(a), n, C.c_uint64(42), C.c_uint64(0), stream())
    call("ddpm_philox_normal", P(b), n, C.c_uint64(42), C.c_uint64(1), stream())
    call("ddpm_philox_normal", P(c), n, C.c_uint64(43), C.c_uint64(0), stream())
    for t in (a, b, c):
        assert torch.isfinite(t).all()
        assert abs(t.mean().item()) < 4.0 / math.sqrt(n) and abs(t.var().item() - 1.0) < 0.005
        assert abs((t ** 3).mean().item()) < 0.02 and abs((t ** 4).mean().item() - 3.0) < 0.05
        assert abs((t[:-1] * t[1:]).mean().item()) < 5.0 / math.sqrt(n)           # neighbours are uncorrelated
        assert t.abs().max().item() < 6.0 and (t.abs() > 3).float().mean().item() == pytest.approx(0.0027, abs=0.0005)
    assert abs((a * b).mean().item()) < 5.0 / math.sqrt(n) and abs((a * c).mean().item()) < 5.0 / math.sqrt(n)
    # element i depends on (seed, call, i) only: a shorter or offset request reproduces the same values
    again = torch.empty(1001, device="cuda")
    call("ddpm_philox_normal", P(again), 1001, C.c_uint64(42), C.c_uint64(0), stream())
    assert torch.equal(again, a[:1001])


def _tiny(variant, seed):
    cin = 1 if variant < 3 else 3
    U = pkg.make_unet_class(variant)
    m = U(cin, cin, 8, 16, num_classes=10) if variant == 3 else U(cin, cin, 8, 16)
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed))
    return m.cuda().eval(), cin


@pytest.mark.parametrize("variant,kw", [(1, {}), (2, {}), (3, {}), (3, dict(labels=torch.tensor([4]), cfg_scale=1.5)), (3, dict(labels=torch.tensor([2, 7])))])
def test_sampling_trajectory_is_bit_identical_with_and_without_the_fused_kernel(variant, kw):
    m, cin = _tiny(variant, variant)
    T, shape = 20, (2, cin, 28, 28)
    betas = pkg.BetaScheduler(T, device="cuda").get_schedule("linear")
    D = pkg.make_diffusion_class(variant)
    xT = seeded_tensor(shape, 5, "xT")
    zs = [seeded_tensor(shape, 5, f"z{i}") for i in range(T - 1)]
    runs = []
    for fuse in (True, False):
        dm = D(m, betas, T)
        dm.fuse_final_step = fuse
        with torch.no_grad():
            runs.append(dm.sample(shape, "cuda", noise=(xT, zs), **kw))
        assert len(m.feature_maps) == 7
    for (a, ta), (b, tb) in zip(*runs):
        assert ta == tb and torch.equal(a, b)


def test_in_kernel_noise_sampling_is_seeded_and_finite():
    m, cin = _tiny(2, 2)
    T, shape = 20, (4, cin, 28, 28)
    dm = pkg.make_diffusion_class(2)(m, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
    outs = []
    for seed in (7, 7, 8):
        torch.manual_seed(seed)
        with torch.no_grad():
            outs.append(dm.sample(shape, "cuda", in_kernel_noise=True))
    assert all(torch.isfinite(s[0]).all() for s in outs[0]) and [s[1] for s in outs[0]] == [18, 16, 14, 12, 10, 8, 6, 4, 2, 0]
    assert all(torch.equal(a[0], b[0]) for a, b in zip(outs[0], outs[1]))          # same seed, same images
    assert not torch.equal(outs[0][-1][0], outs[2][-1][0])                          # another seed, other noise
    # the loop with the same Philox draws injected as tensors gives the same trajectory
    torch.manual_seed(7)
    xT = torch.randn(shape)
    seed = int(torch.randint(0, 2 ** 62, (1,)).item())
    zs = []
    for t in range(T - 1, 0, -1):
        z = torch.empty(shape, device="cuda")
        call("ddpm_philox_normal", P(z), z.numel(), C.c_uint64(seed), C.c_uint64(t), stream())
        zs.append(z.cpu())
    with torch.no_grad():
        inj = dm.sample(shape, "cuda", noise=(xT, zs))
    assert all(torch.equal(a[0], b[0]) for a, b in zip(outs[0], inj))
    m.train()
    with pytest.raises(RuntimeError, match="in_kernel_noise"):
        dm.sample(shape, "cuda", in_kernel_noise=True)


def test_to_images_matches_the_callers_expression_bit_for_bit():
    x = rnd((3, 1, 32, 32), 1, 1.5)
    x.view(-1)[:6] = torch.tensor([float("nan"), -1.0, 1.0, -0.0, 3.0, -3.0])
    got = process.to_images(x.cuda()).cpu()
    want = O.to_images(x)
    assert got.numpy().tobytes() == want.numpy().tobytes()
    with pytest.raises(RuntimeError, match="MI355X"):
        process.to_images(x)


def test_sample_images_returns_unit_range_images_of_the_final_stage():
    m, cin = _tiny(1, 1)
    T, shape = 10, (3, cin, 28, 28)
    dm = pkg.make_diffusion_class(1)(m, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
    xT = seeded_tensor(shape, 5, "xT")
    zs = [seeded_tensor(shape, 5, f"z{i}") for i in range(T - 1)]
    with torch.no_grad():
        imgs = dm.sample_images(shape, "cuda", noise=(xT, zs))
        stages = dm.sample(shape, "cuda", noise=(xT, zs))
    assert not imgs.is_cuda and imgs.min() >= 0 and imgs.max() <= 1
    assert torch.equal(imgs, O.to_images(stages[-1][0]))


@pytest.mark.parametrize("n,c,h,w,nrow,pad,norm", [(10, 1, 32, 32, 5, 2, True), (64, 1, 28, 28, 8, 2, True), (7, 3, 32, 32, 8, 2, False),
                                                   (7, 3, 9, 5, 3, 0, True), (1, 1, 8, 8, 8, 2, True), (1, 3, 8, 8, 8, 2, False),
                                                   (13, 3, 6, 7, 4, 3, True)])
def test_make_grid_matches_the_restated_torchvision_layout(n, c, h, w, nrow, pad, norm):
    x = torch.sigmoid(rnd((n, c, h, w), n + c))
    got = process.make_grid(x.cuda(), nrow=nrow, padding=pad, normalize=norm).cpu()
    want = O.make_grid(x, nrow=nrow, padding=pad, normalize=norm)
    assert got.shape == want.shape
    assert np.abs(got.numpy() - want.numpy()).max() <= (1e-6 if norm else 0.0)
