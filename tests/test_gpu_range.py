"""Operands beyond fp16's range. The default arithmetic represents every fp32 operand as a pair of fp16 numbers, which
holds magnitudes up to 65504; the reference computes in IEEE fp32 (second-iteration/unet.py:16-21, no autocast), which
holds 3e38. These tests pin what happens beyond 65504: every pair-forming kernel raises the library's range flag, and
the host mirror reacts to it -- sample() repeats the steps since its last stage hand-off with three bf16 planes per
operand (full fp32 range) and a training loop moves there from the next step on, each with a RuntimeWarning -- so that
the results are the UNCLIPPED ones, checked here against the fp32 CPU oracle.
"""
import ctypes as C
import warnings

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from gpu_util import ConvDesc, Epilogue, P, assert_close, call, from_nhwc, packed_floats, rnd, stream, to_nhwc
from oracle import ddpm_oracle as O
from seeding import seeded_state_dict, seeded_tensor
import diffusion_model_amd as pkg
from diffusion_model_amd._lib import conv_math, math_scope

pytestmark = pytest.mark.gpu
TOL = 2e-5

pair_default = pytest.mark.skipif(conv_math() != "f16x2", reason="the range watch belongs to the fp16-pair arithmetic")


def read_flag(clear=True):
    host = torch.zeros(1, dtype=torch.int32).pin_memory()
    call("ddpm_range_flag_read", C.c_void_p(host.data_ptr()), int(clear), stream())
    torch.cuda.synchronize()
    return int(host[0])


def conv3x3(x, w, cout):
    B, ci, H, W = x.shape
    wp = torch.empty(packed_floats(9, cout, ci), device="cuda")
    call("ddpm_pack_conv3x3", P(w.cuda().contiguous()), cout, ci, 0, P(wp), stream())
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=cout, ldo=cout, act=0, ldr=0, out_h=H, out_w=W)
    ep = Epilogue(None, None, None, None)
    out = torch.empty((B, H, W, cout), device="cuda")
    call("ddpm_conv3x3_fwd", C.byref(d), P(to_nhwc(x)), None, P(wp), C.byref(ep), P(out), stream())
    return from_nhwc(out)


@pair_default
@pytest.mark.parametrize("cout", [64, 32])
def test_conv3x3_beyond_fp16_range_raises_the_flag_and_bf16x3_gives_the_unclipped_result(cout):
    """Halo kernel (64 output channels) and generic kernel (32): activations of 3e5 and 1e7, as an untrained network's
    samples reach late in a 1000-step loop. The pair arithmetic raises the flag (and only then); the three-plane
    arithmetic reproduces the fp64 convolution of the UNCLIPPED input to the usual tolerance and raises nothing."""
    B, ci, H, W = 2, 32, 12, 12
    x, w = rnd((B, ci, H, W), 31), rnd((cout, ci, 3, 3), 32, 0.1)
    big = x.clone()
    big[0, 5, 3, 3], big[0, 7, 3, 4], big[1, 0, 8, 8] = 3.0e5, -1.0e7, 7.0e4
    read_flag()
    got = conv3x3(x, w, cout)
    assert read_flag() == 0, "operands inside the range must not raise the flag"
    assert_close(got, F.conv2d(x.double(), w.double(), padding=1), TOL, "clean input, fp16 pair")
    clipped = conv3x3(big, w, cout)
    assert torch.isfinite(clipped).all()                       # clipped, not NaN: the flag is what says so
    assert read_flag() == 1, "an operand beyond 65504 must raise the flag"
    assert read_flag() == 0, "reading with clear resets it"
    with math_scope("bf16x3"):
        exact = conv3x3(big, w, cout)
    assert read_flag() == 0
    assert_close(exact, F.conv2d(big.double(), w.double(), padding=1), TOL, "input beyond fp16's range, three bf16 planes")


@pair_default
def test_every_pair_forming_kernel_raises_the_flag():
    """1x1 convolutions (generic, weights resident in LDS, weight slabs), the weight gradient, the attention forward of
    an inference call, and the weight-packing kernels: one operand of 1e5 each."""
    # 1x1: generic (few rows), resident (>= 16384 rows, 64 channels), slab (>= 16384 rows, 256 channels)
    for rows, ci, co in ((256, 64, 80), (16384, 64, 80), (16384, 256, 320)):
        x = rnd((rows, ci), 1)
        w = rnd((co, ci), 2, ci ** -0.5)
        wp = torch.empty(packed_floats(1, co, ci), device="cuda")
        call("ddpm_pack_conv1x1", P(w.cuda()), co, ci, 0, P(wp), stream())
        d = ConvDesc(batch=1, in_h=rows, in_w=1, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=rows, out_w=1)
        ep = Epilogue(None, None, None, None)
        out = torch.empty((rows, co), device="cuda")
        read_flag()
        call("ddpm_conv1x1_fwd", C.byref(d), P(x.cuda()), None, P(wp), C.byref(ep), P(out), stream())
        assert read_flag() == 0, (rows, ci, co)
        assert_close(out.cpu(), x.double() @ w.double().T, TOL, f"1x1 {rows}x{ci}->{co}")
        xb = x.clone()
        xb[rows // 2, 3] = 1.0e5
        call("ddpm_conv1x1_fwd", C.byref(d), P(xb.cuda()), None, P(wp), C.byref(ep), P(out), stream())
        assert read_flag() == 1, (rows, ci, co)
    # weight gradient of a 3x3 layer with 128-wide tiles
    B, H, W, ci, co = 2, 8, 8, 128, 128
    dy, x = rnd((B, H, W, co), 3), rnd((B, H, W, ci), 4)
    packed, gw = torch.empty(co * ci * 9, device="cuda"), torch.empty((co, ci, 3, 3), device="cuda")
    for bad, want in ((False, 0), (True, 1)):
        xx = x.clone()
        if bad:
            xx[1, 2, 3, 4] = -2.0e5
        read_flag()
        call("ddpm_conv3x3_wgrad", P(dy.cuda()), co, co, P(xx.cuda()), ci, ci, B, H, W, P(packed), P(gw), stream())
        assert read_flag() == want, ("wgrad", bad)
    # attention forward, inference call (no log-sum-exp), 128 channels, 16 query channels, 256 positions
    b, n, c, dq = 2, 256, 128, 16
    qkv, xin = rnd((b, n, 2 * dq + c), 5, 0.3), rnd((b, n, c), 6)
    gamma, out = torch.tensor([0.7]).cuda(), torch.empty((b, n, c), device="cuda")
    for bad, want in ((False, 0), (True, 1)):
        q = qkv.clone()
        if bad:
            q[1, 100, 2 * dq + 5] = 9.0e4          # a value channel
        read_flag()
        call("ddpm_attention_fwd", P(q.cuda()), P(xin.cuda()), c, P(gamma), P(out), c, None, None, None, b, n, c, dq, stream())
        assert read_flag() == want, ("attention", bad)
    # packing a weight beyond the range
    w = rnd((64, 64, 3, 3), 7, 0.05)
    wp = torch.empty(packed_floats(9, 64, 64), device="cuda")
    for bad, want in ((False, 0), (True, 1)):
        ww = w.clone()
        if bad:
            ww[3, 4, 1, 1] = 1.0e6
        read_flag()
        call("ddpm_pack_conv3x3", P(ww.cuda()), 64, 64, 0, P(wp), stream())
        assert read_flag() == want, ("pack", bad)


@pair_default
@pytest.mark.parametrize("bad", [float("nan"), float("inf"), float("-inf")])
@pytest.mark.parametrize("cout", [64, 32])
def test_nan_and_infinite_operands_raise_the_flag_and_propagate_on_three_planes(bad, cout):
    """fp32 arithmetic hands a NaN or an infinity on to everything it touches (the reference's loss turns NaN, which is
    how a broken input shows). The fp16 matrix instruction of a wave that saturates its conversions does NOT: a NaN
    operand comes out as finite numbers. So such an operand must raise the flag like one beyond the range, and the
    arithmetic the caller then repeats the work in must propagate it exactly where fp32 does."""
    B, ci, H, W = 2, 32, 12, 12
    x, w = rnd((B, ci, H, W), 41), rnd((cout, ci, 3, 3), 42, 0.1)
    xb = x.clone()
    xb[1, 7, 4, 5] = bad
    read_flag()
    conv3x3(xb, w, cout)
    assert read_flag() == 1, "a NaN / infinite operand must raise the flag"
    with math_scope("bf16x3"):
        got = conv3x3(xb, w, cout)
    # fp32 on the CPU: not finite in the 3x3 neighbourhood, all channels (NaN for a NaN; +-inf for an infinity, which
    # the three planes turn into NaN, inf = inf + (inf - inf): not finite either way, which is what a caller tests)
    want = F.conv2d(xb, w, padding=1)
    assert torch.equal(~torch.isfinite(got.cpu()), ~torch.isfinite(want)) and (~torch.isfinite(want)).sum().item() == 9 * cout
    if bad != bad:
        assert torch.equal(torch.isnan(got.cpu()), torch.isnan(want))
    ok = torch.isfinite(want)
    assert_close(got.cpu()[ok], F.conv2d(x.double(), w.double(), padding=1)[ok], TOL, "elements the bad operand does not reach")


@pair_default
def test_sample_of_a_nan_input_returns_nan_as_the_reference_does():
    """x_T with one NaN pixel: the reference's trajectory is NaN from the first step on (train-free eval BatchNorm does
    not mix samples, so only that sample). The fp16 pair would return finite numbers for it; sample() must not."""
    T, shape = 10, (2, 1, 28, 28)
    m, sd = tiny(2, 2)
    m.eval()
    dm = pkg.make_diffusion_class(2)(m, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
    xT = seeded_tensor(shape, 5, "xT")
    xT[1, 0, 10, 10] = float("nan")
    zs = [seeded_tensor(shape, 5, f"z{i}") for i in range(T - 1)]
    want = oracle_sample(sd, 2, T, xT, zs)
    with pytest.warns(RuntimeWarning, match="beyond"):
        with torch.no_grad():
            got = dm.sample(shape, "cuda", noise=(xT, zs))
    for a, b in zip(got, want):
        bn = torch.as_tensor(b[0])
        assert torch.equal(torch.isnan(a[0]), torch.isnan(bn)), f"NaN pattern at t={a[1]}"
        assert torch.isnan(bn[1]).any() and not torch.isnan(bn[0]).any()
        assert_close(a[0][0], bn[0], 2e-4, f"the clean sample at t={a[1]}")


def tiny(variant, seed):
    U = pkg.make_unet_class(variant)
    m = U(1, 1, 8, 16)
    sd = seeded_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    return m.cuda(), sd


def oracle_sample(sd, variant, T, xT, zs):
    def model_fn(x, tt, labels):
        with torch.no_grad():
            return O.unet_forward(sd, variant, x, tt, labels, train=False)[0]
    return O.sample(model_fn, O.schedule_betas("linear", T), xT.numpy(), [z.numpy() for z in zs])


@pair_default
@pytest.mark.parametrize("blow_up_at", [0, 9])
def test_sample_falls_back_to_full_range_arithmetic_and_matches_the_fp32_oracle(blow_up_at):
    """A trajectory whose activations leave fp16's range -- from the first step on (x_T of 2e5) or from the tenth (one
    injected noise tensor of 1e6) -- must come out as the fp32 reference computes it, not clipped: sample() warns, repeats
    the steps since the last clean stage with three bf16 planes and stays there; the stage list keeps its structure."""
    T, shape = 20, (2, 1, 28, 28)
    m, sd = tiny(2, 2)
    m.eval()
    dm = pkg.make_diffusion_class(2)(m, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
    xT = seeded_tensor(shape, 5, "xT")
    zs = [seeded_tensor(shape, 5, f"z{i}") for i in range(T - 1)]
    if blow_up_at == 0:
        xT = xT * 2.0e5
    else:
        zs[blow_up_at] = zs[blow_up_at] * 1.0e6
    want = oracle_sample(sd, 2, T, xT, zs)
    assert max(s[0].abs().max().item() for s in want) > 1e5            # the case is not vacuous
    with pytest.warns(RuntimeWarning, match="beyond"):
        with torch.no_grad():
            got = dm.sample(shape, "cuda", noise=(xT, zs))
    assert m._engine.math == "bf16x3" and m._engine.range_fallbacks == 1
    assert [s[1] for s in got] == [s[1] for s in want] == [18, 16, 14, 12, 10, 8, 6, 4, 2, 0]
    for a, b in zip(got, want):
        assert_close(a[0], b[0], 2e-4, f"stage t={a[1]}")
    # the same call again: no second warning, the network stays on the full-range arithmetic
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        with torch.no_grad():
            again = dm.sample(shape, "cuda", noise=(xT, zs))
    # which equals, bit for bit, a run of a network that was on that arithmetic from the start
    m2, _ = tiny(2, 2)
    m2.eval()
    m2._engine.math = "bf16x3"
    dm2 = pkg.make_diffusion_class(2)(m2, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
    with torch.no_grad():
        full = dm2.sample(shape, "cuda", noise=(xT, zs))
    assert all(torch.equal(a[0], b[0]) for a, b in zip(again, full))
    if blow_up_at == 0:
        # the range was left in the first step: everything was repeated, the clipped steps left no trace
        assert all(torch.equal(a[0], b[0]) for a, b in zip(got, full))
    else:
        # steps 19 .. 10 had run (unclipped) on the fp16 pair, steps 9 .. 0 were repeated on three bf16 planes
        for a, b in zip(got, full):
            assert_close(a[0], b[0], 2e-4, f"against bf16x3 throughout, t={a[1]}")


@pair_default
def test_sample_with_drawn_noise_repeats_the_same_noise_after_a_fallback():
    """Without injected noise the per-step z comes from torch's GPU generator: a repeat after a fallback restores the
    generator's state, so the call equals one on the full-range arithmetic from the start, given the same seeds."""
    T, shape = 20, (2, 1, 28, 28)
    outs = []
    for start_full in (False, True):
        m, sd = tiny(2, 2)
        with torch.no_grad():
            sd2 = {k: v.clone() for k, v in m.state_dict().items()}
            sd2["enc2.conv.double_conv.1.weight"] *= 3.0e5           # a BatchNorm gain that drives activations to 1e5
            sd2["final_conv.weight"] *= 1.0e-6                       # (and an output layer that keeps the trajectory finite)
            sd2["final_conv.bias"] *= 1.0e-6
            m.load_state_dict(sd2)
        m.eval()
        if start_full:
            m._engine.math = "bf16x3"
        dm = pkg.make_diffusion_class(2)(m, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
        torch.manual_seed(11)
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            with torch.no_grad():
                outs.append(dm.sample(shape, "cuda"))
        assert bool([w for w in rec if issubclass(w.category, RuntimeWarning)]) == (not start_full)
        assert m._engine.math == "bf16x3"
        assert all(torch.isfinite(s[0]).all() for s in outs[-1])
    assert all(torch.equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(*outs))


@pair_default
def test_training_moves_to_full_range_arithmetic_after_a_step_that_left_the_range():
    """A BatchNorm gain of 4e5 puts the next convolution's input beyond fp16's range. The step that meets it is
    computed on clipped operands (nothing waits for the GPU inside a step); the next call finds the flag, warns and
    computes with three bf16 planes: its loss and gradients are the fp32 oracle's."""
    m, sd = tiny(2, 6)
    sd = {k: v.clone() for k, v in sd.items()}
    sd["enc1.double_conv.1.weight"] = sd["enc1.double_conv.1.weight"] * 4.0e5
    m.load_state_dict(sd)
    m.train()
    x, t = seeded_tensor((4, 1, 28, 28), 1, "x"), torch.tensor([5, 900, 17, 0])
    tgt = seeded_tensor((4, 1, 28, 28), 2, "tgt")
    crit = torch.nn.MSELoss()
    crit(m(x.cuda(), t.cuda()), tgt.cuda()).backward()
    torch.cuda.synchronize()
    assert m._engine.math is None
    m.zero_grad()
    bufs = {k: v.clone() for k, v in m.state_dict().items() if not O.is_param(k)}   # running statistics after step 1
    with pytest.warns(RuntimeWarning, match="beyond"):
        loss = crit(m(x.cuda(), t.cuda()), tgt.cuda())
    loss.backward()
    assert m._engine.math == "bf16x3"
    cur = dict(sd)
    cur.update({k: v.cpu() for k, v in bufs.items()})
    want_loss, grads, _, _ = O.loss_and_grads(cur, 2, x, t, tgt)
    assert abs(loss.item() - float(want_loss)) <= 1e-4 * abs(float(want_loss))
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:5]


# ---------------------------------------------------------------------------------------------------------------
# Where several workgroups of a launch stage the same operand, only one of them checks it (the first column tile,
# query block, key block or tap group). These tests put ONE out-of-range element at many places of every operand --
# first and last rows, samples, channels, chunk boundaries -- and require the flag each time: no element may fall
# between the workgroups that do the checking.
def _positions(shape, n, seed):
    g = torch.Generator().manual_seed(seed)
    corners = [tuple(0 for _ in shape), tuple(s - 1 for s in shape)]
    rand = [tuple(int(torch.randint(0, s, (1,), generator=g)) for s in shape) for _ in range(n)]
    return corners + rand


@pair_default
@pytest.mark.parametrize("ci,co,hw,batch", [(128, 128, 16, 40), (64, 256, 8, 70), (256, 64, 12, 33), (64, 64, 28, 24)])
def test_conv3x3_and_its_weight_gradient_flag_a_bad_element_anywhere(ci, co, hw, batch):
    x = torch.randn((batch, hw, hw, ci), device="cuda")
    dy = torch.randn((batch, hw, hw, co), device="cuda")
    w = torch.randn((co, ci, 3, 3), device="cuda") * 0.05
    wp = torch.empty(packed_floats(9, co, ci), device="cuda")
    call("ddpm_pack_conv3x3", P(w), co, ci, 0, P(wp), stream())
    out = torch.empty((batch, hw, hw, co), device="cuda")
    d = ConvDesc(batch=batch, in_h=hw, in_w=hw, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=hw, out_w=hw)
    ep = Epilogue(None, None, None, None)
    packed, gw = torch.empty(co * ci * 9, device="cuda"), torch.empty((co, ci, 3, 3), device="cuda")

    def fwd():
        call("ddpm_conv3x3_fwd", C.byref(d), P(x), None, P(wp), C.byref(ep), P(out), stream())

    def wgrad():
        call("ddpm_conv3x3_wgrad", P(dy), co, co, P(x), ci, ci, batch, hw, hw, P(packed), P(gw), stream())

    read_flag()
    fwd()
    wgrad()
    assert read_flag() == 0
    for pos in _positions(x.shape, 6, 1):
        keep = x[pos].item()
        x[pos] = 1.0e5
        for what, run in (("forward", fwd), ("weight gradient, X", wgrad)):
            run()
            assert read_flag() == 1, (what, pos)
        x[pos] = keep
    for pos in _positions(dy.shape, 6, 2):
        keep = dy[pos].item()
        dy[pos] = -3.0e5
        wgrad()
        assert read_flag() == 1, ("weight gradient, dY", pos)
        dy[pos] = keep
    fwd()
    wgrad()
    assert read_flag() == 0


@pair_default
@pytest.mark.parametrize("rows,ci,co", [(20000, 128, 160), (17000, 256, 320), (16500, 512, 640), (3000, 128, 160)])
def test_conv1x1_flags_a_bad_element_anywhere(rows, ci, co):
    x = torch.randn((rows, ci), device="cuda")
    w = torch.randn((co, ci), device="cuda") * ci ** -0.5
    wp = torch.empty(packed_floats(1, co, ci), device="cuda")
    call("ddpm_pack_conv1x1", P(w), co, ci, 0, P(wp), stream())
    d = ConvDesc(batch=1, in_h=rows, in_w=1, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=rows, out_w=1)
    ep = Epilogue(None, None, None, None)
    out = torch.empty((rows, co), device="cuda")
    read_flag()
    for pos in _positions(x.shape, 8, 3):
        keep = x[pos].item()
        x[pos] = 7.0e4
        call("ddpm_conv1x1_fwd", C.byref(d), P(x), None, P(wp), C.byref(ep), P(out), stream())
        assert read_flag() == 1, pos
        x[pos] = keep
    call("ddpm_conv1x1_fwd", C.byref(d), P(x), None, P(wp), C.byref(ep), P(out), stream())
    assert read_flag() == 0


@pair_default
@pytest.mark.parametrize("b,n,c,dq", [(3, 784, 128, 16), (2, 1024, 64, 8), (5, 200, 128, 16)])
def test_attention_forward_and_backward_flag_a_bad_element_anywhere(b, n, c, dq):
    qkv = torch.randn((b, n, 2 * dq + c), device="cuda") * 0.3
    xin, dy = torch.randn((b, n, c), device="cuda"), torch.randn((b, n, c), device="cuda")
    gamma = torch.tensor([0.7], device="cuda")
    out, raw, lse = torch.empty((b, n, c), device="cuda"), torch.empty((b, n, c), device="cuda"), torch.empty((b, n), device="cuda")
    dqkv, dgamma = torch.empty_like(qkv), torch.empty(1, device="cuda")
    ws, es = torch.empty(4096, dtype=torch.float64, device="cuda"), torch.empty((b, n), device="cuda")

    def fwd(train):
        call("ddpm_attention_fwd", P(qkv), P(xin), c, P(gamma), P(out), c, P(raw) if train else None, P(lse) if train else None, None,
             b, n, c, dq, stream())

    def bwd():
        call("ddpm_attention_bwd", P(qkv), P(dy), c, P(raw), P(lse), P(gamma), P(ws), P(es), P(dqkv), P(dgamma), b, n, c, dq, stream())

    read_flag()
    fwd(True)
    bwd()
    fwd(False)
    assert read_flag() == 0
    for pos in _positions(qkv.shape, 8, 4):
        keep = qkv[pos].item()
        fwd(True)                      # a clean forward: the backward below sees the bad element in qkv only
        read_flag()
        qkv[pos] = 9.0e4
        bwd()
        assert read_flag() == 1, ("backward, q|k|v", pos)
        for train in (False, True):
            fwd(train)
            assert read_flag() == 1, ("forward", train, pos)
        qkv[pos] = keep
    fwd(True)
    read_flag()
    for pos in _positions(dy.shape, 8, 5):
        keep = dy[pos].item()
        dy[pos] = -2.0e5
        bwd()
        assert read_flag() == 1, ("backward, dY", pos)
        dy[pos] = keep
    bwd()
    assert read_flag() == 0


@pair_default
@pytest.mark.parametrize("b,n,c", [(16, 1024, 128), (64, 256, 64), (3, 256, 128)])
def test_self_attention_infer_flags_what_its_projection_splits(b, n, c):
    """The fused inference entry splits k and v in the projection's epilogue (first two shapes; the third runs the two plain
    calls): a q, k or v column pushed beyond fp16's range through its bias, or one input element, must raise the flag."""
    dq = c // 8
    dqp = (dq + 7) // 8 * 8
    lq = 2 * dqp + c
    x = torch.randn((b, n, c), device="cuda")
    ws_ = [torch.randn((r, c), device="cuda") * c ** -0.5 for r in (dq, dq, c)]
    bs_ = [torch.zeros(r, device="cuda") for r in (dq, dq, c)]
    wf, wt = torch.empty(packed_floats(1, lq, c), device="cuda"), torch.empty(packed_floats(1, c, lq), device="cuda")
    bias, gamma = torch.empty(lq, device="cuda"), torch.tensor([0.5], device="cuda")
    call("ddpm_pack_qkv", *(P(t) for t in ws_), *(P(t) for t in bs_), c, dq, dqp, P(wf), P(wt), P(bias), stream())
    out, ws = torch.empty((b, n, c), device="cuda"), torch.empty((b, n, lq), device="cuda")

    def run():
        call("ddpm_self_attention_infer", P(x), c, P(wf), P(bias), P(gamma), None, P(out), c, P(ws), b, n, c, dqp, stream())

    read_flag()
    run()
    assert read_flag() == 0
    for col in (0, dq - 1, dqp, dqp + dq - 1, 2 * dqp, 2 * dqp + 17, lq - 1):
        keep = bias[col].item()
        bias[col] = 1.0e5
        run()
        assert read_flag() == 1, ("bias of column", col)
        bias[col] = keep
    for pos in _positions(x.shape, 4, 6):
        keep = x[pos].item()
        x[pos] = -8.0e4
        run()
        assert read_flag() == 1, ("input", pos)
        x[pos] = keep
    run()
    assert read_flag() == 0


@pair_default
@pytest.mark.parametrize("fuse", [True, False])
def test_guided_sampling_falls_back_as_a_whole_segment(fuse):
    """Classifier-free guidance calls the network twice per step (third-iteration/diffusion.py:34-41), once through the
    plain forward: that call must leave the flag to sample(), which repeats the whole segment -- the result equals a run
    on three bf16 planes from the start, bit for bit, with and without the fused last kernel."""
    T, shape = 10, (2, 1, 28, 28)
    xT = seeded_tensor(shape, 5, "xT") * 3.0e5
    zs = [seeded_tensor(shape, 5, f"z{i}") for i in range(T - 1)]
    labels = torch.tensor([3])
    outs = []
    for start_full in (False, True):
        m = pkg.make_unet_class(3)(1, 1, 8, 16, num_classes=10)
        m.load_state_dict(seeded_state_dict(m.state_dict(), 4))
        m = m.cuda().eval()
        if start_full:
            m._engine.math = "bf16x3"
        cls = pkg.make_diffusion_class(3)
        dm = cls(m, pkg.BetaScheduler(T, device="cuda").get_schedule("linear"), T)
        dm.fuse_final_step = fuse
        with warnings.catch_warnings(record=True) as rec:
            warnings.simplefilter("always")
            with torch.no_grad():
                outs.append(dm.sample(shape, "cuda", labels=labels, cfg_scale=2.0, noise=(xT, zs)))
        assert len([w for w in rec if issubclass(w.category, RuntimeWarning)]) == (0 if start_full else 1)
        assert m._engine.math == "bf16x3" and m._engine.range_fallbacks == (0 if start_full else 1)
        assert not m._engine._range_external
    assert all(torch.equal(a[0], b[0]) and a[1] == b[1] for a, b in zip(*outs))
    assert max(s[0].abs().max().item() for s in outs[0]) > 1e5
