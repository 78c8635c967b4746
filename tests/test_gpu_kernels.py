"""Per-kernel parity on the GPU, through the C ABI, against the CPU oracle / torch fp32 on CPU.

Tolerances: elementwise process kernels are bit-exact; MFMA contractions are fp32 with a
different summation order than the CPU, so they are held to 2e-5 of the tensor's scale
(K up to 9*512 products per output).
"""
import ctypes as C

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden
from gpu_util import ConvDesc, Epilogue, P, assert_close, call, from_nhwc, packed_floats, rnd, stream, to_nhwc
from oracle import ddpm_oracle as O

pytestmark = pytest.mark.gpu
TOL = 2e-5


# ---------------------------------------------------------------- elementwise, bit-exact
@pytest.mark.parametrize("tag", ["a", "b"])
def test_reverse_step_bit_exact_golden(tag):
    g = load_golden("elementwise.npz")
    x, eps, z = (torch.from_numpy(g[f"{tag}_{k}"]).cuda() for k in ("x0", "eps", "z"))
    for tt in (999, 500, 1, 0):
        c1, c2, c3 = (float(v) for v in g[f"{tag}_coef_t{tt}"])
        out = torch.empty_like(x)
        call("ddpm_reverse_step", P(x), P(eps), None, 0.0, P(z) if tt > 0 else None, c1, c2, c3, P(out), x.numel(), stream())
        assert out.cpu().numpy().tobytes() == g[f"{tag}_step_t{tt}"].tobytes()


@pytest.mark.parametrize("n", [1, 3, 4, 1023, 4096 * 3 + 2, 1 << 22, (1 << 23) + 4096 * 5 + 3])   # the last: streaming variant
@pytest.mark.parametrize("w", [0.0, 0.25, 0.5, 2.0])
def test_reverse_step_ragged_and_guided(n, w):
    x, e, u, z = (rnd((n,), s) for s in (1, 2, 3, 4))
    c1, c2, c3 = np.float32(1.0101), np.float32(0.0123), np.float32(0.1414)
    xe, ee, ue, ze = (t.cuda() for t in (x, e, u, z))
    out = torch.empty_like(xe)
    call("ddpm_reverse_step", P(xe), P(ee), P(ue) if w > 0 else None, float(w), P(ze), float(c1), float(c2), float(c3), P(out), n,
         stream())
    eps = O.cfg_blend(u.numpy(), e.numpy(), w) if w > 0 else e.numpy()
    want = O.reverse_step(x.numpy(), eps, z.numpy(), c1, c2, c3)
    assert out.cpu().numpy().tobytes() == want.tobytes()


def test_cfg_blend_golden():
    g = load_golden("elementwise.npz")
    u, c = torch.from_numpy(g["cfg_u"]).cuda(), torch.from_numpy(g["cfg_c"]).cuda()
    x = torch.zeros_like(u)
    for w in (0.25, 0.5, 2.0):
        out = torch.empty_like(u)
        # x = 0, c1 = 1, c2 = -1... use c2 = 1 and negate: out = 0 - 1*eps -> compare -out with lerp
        call("ddpm_reverse_step", P(x), P(c), P(u), float(w), None, 1.0, 1.0, 0.0, P(out), u.numel(), stream())
        assert np.array_equal(-out.cpu().numpy(), g[f"cfg_w{w}"])


@pytest.mark.parametrize("tag", ["a", "b"])
def test_add_noise(tag):
    g = load_golden("elementwise.npz")
    betas = O.schedule_betas("linear", 1000)
    _, abar = O.schedule_tables(betas)
    x0, eps = torch.from_numpy(g[f"{tag}_x0"]).cuda(), torch.from_numpy(g[f"{tag}_eps"]).cuda()
    t = torch.from_numpy(g[f"{tag}_t"]).cuda()
    out = torch.empty_like(x0)
    call("ddpm_add_noise", P(x0), P(eps), P(t), P(torch.from_numpy(abar).cuda()), 1000, P(out), x0.shape[0],
         x0.numel() // x0.shape[0], stream())
    want = O.add_noise(g[f"{tag}_x0"], g[f"{tag}_eps"], g[f"{tag}_t"], abar)
    assert out.cpu().numpy().tobytes() == want.tobytes()                        # bit-exact vs the IEEE oracle
    np.testing.assert_allclose(out.cpu().numpy(), g[f"{tag}_noisy"], rtol=3e-7, atol=1e-7)   # reference: torch sqrt, 1 ulp


def test_add_noise_ragged_sample_size():
    betas = O.schedule_betas("cosine", 100)
    _, abar = O.schedule_tables(betas)
    x0, eps = rnd((5, 3, 7, 5), 1), rnd((5, 3, 7, 5), 2)
    t = torch.tensor([0, 99, 50, 1, 98])
    out = torch.empty((5, 3, 7, 5), device="cuda")
    call("ddpm_add_noise", P(x0.cuda()), P(eps.cuda()), P(t.cuda()), P(torch.from_numpy(abar).cuda()), 100, P(out), 5, 105, stream())
    assert out.cpu().numpy().tobytes() == O.add_noise(x0.numpy(), eps.numpy(), t.numpy(), abar).tobytes()


def test_mse_fwd_bwd():
    n = 3 * 1 * 28 * 28 + 5
    p, y = rnd((n,), 1), rnd((n,), 2)
    pc, yc = p.cuda(), y.cuda()
    part = torch.empty(1024, dtype=torch.float64, device="cuda")
    loss = torch.empty(1, device="cuda")
    call("ddpm_mse_fwd", P(pc), P(yc), n, P(part), P(loss), stream())
    want = F.mse_loss(p.double(), y.double())
    assert abs(loss.item() - want.item()) <= 1e-6 * want.item()
    gout = torch.tensor([0.7], device="cuda")
    grad = torch.empty_like(pc)
    call("ddpm_mse_bwd", P(pc), P(yc), P(gout), n, P(grad), stream())
    assert_close(grad.cpu(), 0.7 * 2 * (p - y) / n, 1e-6, "mse grad")


# ---------------------------------------------------------------- embedding
def test_sinusoidal_embedding_and_linears():
    g = load_golden("embeddings.npz")
    t = torch.arange(1000).cuda()
    out = torch.empty((1000, 128), device="cuda")
    call("ddpm_sinusoidal_embedding", P(t), 0, 1000, 128, P(out), stream())
    # E=128: the frequency table matches the reference bit for bit; sin/cos implementations differ by <= 2 ulp
    np.testing.assert_allclose(out.cpu().numpy(), g["E128_t0_999"], rtol=0, atol=5e-7)
    few = torch.from_numpy(g["t_few"]).cuda()
    for dim in (16, 64, 256):
        o = torch.empty((4, dim), device="cuda")
        call("ddpm_sinusoidal_embedding", P(few), 0, 4, dim, P(o), stream())
        # a 1-ulp frequency difference is amplified by t <= 999 before sin/cos
        np.testing.assert_allclose(o.cpu().numpy(), g[f"E{dim}_few"], rtol=0, atol=1.5e-4)
    o = torch.empty((4, 128), device="cuda")
    call("ddpm_sinusoidal_embedding", P(few.float()), 1, 4, 128, P(o), stream())
    np.testing.assert_allclose(o.cpu().numpy(), g["E128_float_few"], rtol=0, atol=5e-7)


# ---------------------------------------------------------------- convolutions
def conv_ref(x, w, b, scale, shift, res, temb, act):
    y = F.conv2d(x, w, None, padding=w.shape[-1] // 2)
    if scale is not None:
        y = y * scale[None, :, None, None] + shift[None, :, None, None]
    elif shift is not None:
        y = y + shift[None, :, None, None]
    if res is not None:
        y = y + res
    y = F.relu(y) if act == 1 else (F.gelu(y) if act == 2 else y)
    if temb is not None:
        y = y + temb[:, :, None, None]
    return y


def pack3(w):
    out = torch.empty(packed_floats(9, w.shape[0], w.shape[1]), device="cuda")
    call("ddpm_pack_conv3x3", P(w.cuda().contiguous()), w.shape[0], w.shape[1], 0, P(out), stream())
    return out


CONV_CASES = [
    # B, H, W, Cin, Cout, act, scale, residual, temb
    (2, 8, 8, 8, 8, 0, False, False, False),
    (3, 7, 7, 64, 128, 1, True, False, False),
    (2, 28, 28, 64, 64, 2, True, True, True),
    (1, 30, 30, 128, 256, 2, True, True, False),
    (2, 14, 14, 256, 128, 1, False, False, True),
    (2, 7, 7, 256, 512, 0, False, False, False),
    (1, 9, 5, 16, 40, 2, True, False, True),
    (5, 16, 16, 512, 512, 1, True, False, False),
]


@pytest.mark.parametrize("case", CONV_CASES)
def test_conv3x3(case):
    B, H, W, ci, co, act, use_scale, use_res, use_temb = case
    x, w = rnd((B, ci, H, W), 1), rnd((co, ci, 3, 3), 2, (9 * ci) ** -0.5)
    shift = rnd((co,), 3, 0.1)
    scale = rnd((co,), 4).abs() + 0.5 if use_scale else None
    res = rnd((B, co, H, W), 5) if use_res else None
    temb = rnd((B, co), 6) if use_temb else None
    xd = to_nhwc(x)
    out = torch.full((B, H, W, co), float("nan"), device="cuda")
    resd = to_nhwc(res) if use_res else None
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=act, ldr=co, out_h=H, out_w=W)
    keep = [scale.cuda() if use_scale else None, shift.cuda(), temb.cuda() if use_temb else None]
    ep = Epilogue(P(keep[0]), P(keep[1]), P(resd), P(keep[2]))
    call("ddpm_conv3x3_fwd", C.byref(d), P(xd), None, P(pack3(w)), C.byref(ep), P(out), stream())
    assert_close(from_nhwc(out), conv_ref(x, w, None, scale, shift, res, temb, act), TOL, f"conv3x3 {case}")


@pytest.mark.parametrize("case", [(2, 28, 28, 64, 64), (3, 14, 14, 128, 96), (1, 30, 30, 64, 128), (5, 7, 7, 256, 512), (2, 9, 5, 16, 40),
                                  (2, 8, 8, 8, 32), (300, 8, 8, 64, 64)])
def test_conv3x3_with_batchnorm_statistics(case):
    """ddpm_conv3x3_fwd_stats: the same output as ddpm_conv3x3_fwd, bit for bit, and batch statistics finished from the
    partial sums its epilogue leaves that equal ddpm_bn_stats over that output (fp32 sums over 16 values, then float64,
    against float64 throughout: 1e-6 of the value), ragged last tiles, a partly empty column block and the generic
    kernel's shapes (32 output channels) included; running statistics follow momentum 0.1 with the unbiased variance."""
    B, H, W, ci, co = case
    x, w, bias = rnd((B, ci, H, W), 21), rnd((co, ci, 3, 3), 22, (9 * ci) ** -0.5), rnd((co,), 23) + 0.5
    xd, wd, bd = to_nhwc(x), w.cuda().contiguous(), bias.cuda()
    wp = torch.empty(packed_floats(9, co, ci), device="cuda")
    call("ddpm_pack_conv3x3", P(wd), co, ci, 0, P(wp), stream())
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=H, out_w=W)
    ep = Epilogue(None, P(bd), None, None)
    ref_out = torch.empty((B, H, W, co), device="cuda")
    call("ddpm_conv3x3_fwd", C.byref(d), P(xd), None, P(wp), C.byref(ep), P(ref_out), stream())
    out = torch.full((B, H, W, co), float("nan"), device="cuda")
    from diffusion_model_amd._lib import lib
    ws = torch.full((int(lib().ddpm_reduce_workspace_doubles(max(co, 64))),), float("nan"), dtype=torch.float64, device="cuda")
    nblk = C.c_int(0)
    call("ddpm_conv3x3_fwd_stats", C.byref(d), P(xd), None, P(wp), C.byref(ep), P(out), P(ws), C.byref(nblk), stream())
    assert torch.equal(out, ref_out)
    assert 0 < nblk.value <= 1024
    rows = B * H * W
    got = [torch.empty(co, device="cuda") for _ in range(2)] + [torch.full((co,), 0.25, device="cuda"), torch.full((co,), 2.0, device="cuda")]
    call("ddpm_bn_stats_from_partials", P(ws), nblk.value, rows, co, 1e-5, 0.1, *[P(t) for t in got], stream())
    want = [torch.empty(co, device="cuda") for _ in range(2)] + [torch.full((co,), 0.25, device="cuda"), torch.full((co,), 2.0, device="cuda")]
    ws2 = torch.empty_like(ws)
    call("ddpm_bn_stats", P(ref_out), co, rows, co, 1e-5, 0.1, P(ws2), *[P(t) for t in want], stream())
    for g, wnt, name in zip(got, want, ("mean", "invstd", "running_mean", "running_var")):
        assert_close(g.cpu(), wnt.cpu(), 1e-6, f"{name} {case}")
    flat = from_nhwc(ref_out).double().transpose(0, 1).reshape(co, -1)
    assert_close(got[0].cpu(), flat.mean(1), 2e-6, "mean against torch")
    assert_close(got[1].cpu(), 1.0 / torch.sqrt(flat.var(1, unbiased=False) + 1e-5), 2e-6, "invstd against torch")


def test_conv3x3_two_sources_and_strided_output():
    """Input = channel concat of two tensors with their own pixel strides; output into a wider buffer."""
    B, H, W, c1, c2, co = 2, 12, 10, 24, 40, 48
    xa, xb, w = rnd((B, c1, H, W), 1), rnd((B, c2, H, W), 2), rnd((co, c1 + c2, 3, 3), 3, 0.05)
    bufa = torch.zeros((B, H, W, c1 + 8), device="cuda")
    bufa[..., :c1] = to_nhwc(xa)
    bufb = torch.zeros((B, H, W, c2 + 4), device="cuda")
    bufb[..., 4:] = to_nhwc(xb)
    out = torch.full((B, H, W, co + 16), 7.0, device="cuda")
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=c1, ld1=c1 + 8, c2=c2, ld2=c2 + 4, cout=co, ldo=co + 16, act=0, ldr=0, out_h=H, out_w=W)
    ep = Epilogue(None, None, None, None)
    call("ddpm_conv3x3_fwd", C.byref(d), P(bufa), C.c_void_p(bufb.data_ptr() + 16), P(pack3(w)), C.byref(ep),
         C.c_void_p(out.data_ptr() + 32), stream())
    want = F.conv2d(torch.cat([xa, xb], 1), w, padding=1)
    assert_close(from_nhwc(out[..., 8:8 + co]), want, TOL, "dual-source conv")
    assert torch.all(out[..., :8] == 7.0) and torch.all(out[..., 8 + co:] == 7.0)      # nothing outside the slice


@pytest.mark.parametrize("case", [(2, 49, 64, 80), (3, 100, 8, 24), (1, 1024, 128, 160), (2, 64, 512, 640)])
def test_conv1x1(case):
    B, N, ci, co = case
    x, w, b = rnd((B, ci, N, 1), 1), rnd((co, ci), 2, ci ** -0.5), rnd((co,), 3)
    out = torch.empty((B, N, 1, co), device="cuda")
    d = ConvDesc(batch=B, in_h=N, in_w=1, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=N, out_w=1)
    bd, wd = b.cuda(), torch.empty(packed_floats(1, co, ci), device="cuda")
    call("ddpm_pack_conv1x1", P(w.cuda()), co, ci, 0, P(wd), stream())
    ep = Epilogue(None, P(bd), None, None)
    call("ddpm_conv1x1_fwd", C.byref(d), P(to_nhwc(x)), None, P(wd), C.byref(ep), P(out), stream())
    assert_close(from_nhwc(out), F.conv2d(x, w[:, :, None, None], b), TOL, f"conv1x1 {case}")


@pytest.mark.parametrize("case", [(32, 784, 128, 160, 0), (24, 1024, 64, 80, 0), (21, 784, 160, 128, 1), (17, 1024, 80, 64, 1), (40, 441, 128, 72, 2),
                                  (33, 500, 68, 192, 0), (20, 900, 128, 33, 2),
                                  # weights too large for LDS: column tiles and K slabs (ragged last tile, a single slab, residual, GELU)
                                  (40, 441, 256, 320, 0), (36, 500, 512, 640, 1), (33, 500, 256, 256, 2), (34, 490, 320, 200, 0),
                                  (35, 480, 64, 448, 2), (20, 900, 192, 224, 1)])
def test_conv1x1_many_rows(case):
    """Shapes of the attention projections at full resolution (rows >= 16384, weights small enough to stay in LDS): the
    kernel with resident weights. mode 0: bias; 1: the data-gradient form (no bias, residual add, strided residual);
    2: scale/shift + GELU. Ragged row counts, columns that do not fill their last 32-block, K that is not a multiple of 16."""
    B, N, ci, co, mode = case
    x, w = rnd((B, ci, N, 1), 1), rnd((co, ci), 2, ci ** -0.5)
    b, sc = rnd((co,), 3), rnd((co,), 4).abs() + 0.5
    res = rnd((B, co, N, 1), 5)
    out = torch.full((B, N, 1, co + 4), 7.0, device="cuda")
    wd = torch.empty(packed_floats(1, co, ci), device="cuda")
    call("ddpm_pack_conv1x1", P(w.cuda()), co, ci, 0, P(wd), stream())
    resd = torch.zeros((B, N, 1, co + 8), device="cuda")
    resd[..., :co] = to_nhwc(res)
    d = ConvDesc(batch=B, in_h=N, in_w=1, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co + 4, act=2 if mode == 2 else 0, ldr=co + 8, out_h=N, out_w=1)
    keep = [b.cuda(), sc.cuda()]
    ep = {0: Epilogue(None, P(keep[0]), None, None), 1: Epilogue(None, None, P(resd), None), 2: Epilogue(P(keep[1]), P(keep[0]), None, None)}[mode]
    call("ddpm_conv1x1_fwd", C.byref(d), P(to_nhwc(x)), None, P(wd), C.byref(ep), P(out), stream())
    want = F.conv2d(x, w[:, :, None, None])
    want = {0: want + b[None, :, None, None], 1: want + res, 2: F.gelu(want * sc[None, :, None, None] + b[None, :, None, None])}[mode]
    assert_close(from_nhwc(out[..., :co]), want, TOL, f"conv1x1 many rows {case}")
    assert torch.all(out[..., co:] == 7.0)
    # the same rows in pieces of fewer than 16384 rows take the generic GEMM kernel: identical bits, so that a result does
    # not depend on how a batch was cut
    xd = to_nhwc(x)
    pieces = []
    for b0 in range(0, B, 8):
        nb = min(8, B - b0)
        o = torch.full((nb, N, 1, co + 4), 7.0, device="cuda")
        d2 = ConvDesc(batch=nb, in_h=N, in_w=1, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co + 4, act=d.act, ldr=co + 8, out_h=N, out_w=1)
        rp = resd[b0:b0 + nb]
        ep2 = {0: Epilogue(None, P(keep[0]), None, None), 1: Epilogue(None, None, P(rp), None), 2: Epilogue(P(keep[1]), P(keep[0]), None, None)}[mode]
        call("ddpm_conv1x1_fwd", C.byref(d2), P(xd[b0:b0 + nb]), None, P(wd), C.byref(ep2), P(o), stream())
        pieces.append(o)
    assert torch.equal(torch.cat(pieces), out)


@pytest.mark.parametrize("case", [(2, 7, 7, 64, 32, 14, 14), (2, 7, 7, 32, 16, 15, 15), (2, 16, 16, 32, 16, 16, 16),
                                  (1, 14, 14, 128, 64, 28, 28), (3, 15, 15, 8, 4, 15, 15), (2, 7, 5, 16, 8, 17, 9),
                                  # output channels in whole 32-column blocks (the table-driven scatter): padded, cropped to
                                  # half (the reference's third up block), cropped by odd amounts, ragged last tile
                                  (2, 7, 7, 64, 32, 15, 15), (2, 16, 16, 32, 64, 16, 16), (2, 7, 5, 16, 32, 17, 9),
                                  (1, 9, 9, 32, 32, 13, 15), (5, 11, 13, 64, 96, 19, 23), (3, 32, 32, 128, 64, 32, 32)])
def test_convT2x2_pad_crop(case):
    B, H, W, ci, co, OH, OW = case
    x, w, b = rnd((B, ci, H, W), 1), rnd((ci, co, 2, 2), 2, ci ** -0.5), rnd((co,), 3)
    y = F.conv_transpose2d(x, w, b, stride=2)
    dy, dx = OH - y.shape[2], OW - y.shape[3]
    want = F.pad(y, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])
    wp = torch.empty(packed_floats(1, 4 * co, ci), device="cuda")
    call("ddpm_pack_convT2x2", P(w.cuda()), ci, co, P(wp), stream())
    out = torch.zeros((B, OH, OW, co), device="cuda")
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=OH, out_w=OW,
                 off_y=dy // 2, off_x=dx // 2)
    bd = b.cuda()
    ep = Epilogue(None, P(bd), None, None)
    call("ddpm_convT2x2_fwd", C.byref(d), P(to_nhwc(x)), P(wp), C.byref(ep), P(out), stream())
    assert_close(from_nhwc(out), want, TOL, f"convT {case}")


@pytest.mark.parametrize("case", [(80, 16, 16, 256, 128), (70, 16, 16, 64, 40), (21, 32, 32, 128, 64), (67, 16, 16, 192, 72), (130, 11, 13, 64, 64)])
def test_convT2x2_many_rows(case):
    """16384 input pixels and more with the output exactly twice the input (the shapes of the sampling and training
    batches). Against torch, with a wider output row than cout (ldo), untouched padding columns, and bit for bit against
    the same samples in small batches: a result must not depend on how a batch was cut."""
    B, H, W, ci, co = case
    x, w, b = rnd((B, ci, H, W), 1), rnd((ci, co, 2, 2), 2, ci ** -0.5), rnd((co,), 3)
    want = F.conv_transpose2d(x, w, b, stride=2)
    wp = torch.empty(packed_floats(1, 4 * co, ci), device="cuda")
    call("ddpm_pack_convT2x2", P(w.cuda()), ci, co, P(wp), stream())
    ldo = co + 8
    out = torch.full((B, 2 * H, 2 * W, ldo), 7.0, device="cuda")
    xd, bd = to_nhwc(x), b.cuda()
    ep = Epilogue(None, P(bd), None, None)
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=ldo, act=0, ldr=0, out_h=2 * H, out_w=2 * W)
    call("ddpm_convT2x2_fwd", C.byref(d), P(xd), P(wp), C.byref(ep), P(out), stream())
    assert_close(from_nhwc(out[..., :co].contiguous()), want, TOL, f"convT, many rows {case}")
    assert torch.all(out[..., co:] == 7.0)
    cut = 8
    small = torch.empty((cut, 2 * H, 2 * W, ldo), device="cuda")
    for b0 in (0, B - cut):
        d2 = ConvDesc(batch=cut, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=ldo, act=0, ldr=0, out_h=2 * H, out_w=2 * W)
        call("ddpm_convT2x2_fwd", C.byref(d2), P(xd[b0:b0 + cut]), P(wp), C.byref(ep), P(small), stream())
        assert torch.equal(small[..., :co], out[b0:b0 + cut, ..., :co]), f"convT {case}: batch of {B} against batch of {cut} at sample {b0}"


@pytest.mark.parametrize("cin", [1, 3])
def test_first_and_final_conv(cin):
    B, H, W, co = 3, 11, 13, 16
    x, w = rnd((B, cin, H, W), 1), rnd((co, cin, 3, 3), 2, 0.3)
    scale, shift, temb = rnd((co,), 3).abs() + 0.5, rnd((co,), 4), rnd((B, co), 5)
    out = torch.empty((B, H, W, co), device="cuda")
    keep = [scale.cuda(), shift.cuda(), temb.cuda(), w.cuda()]
    ep = Epilogue(P(keep[0]), P(keep[1]), None, P(keep[2]))
    call("ddpm_conv3x3_first_fwd", P(x.cuda()), P(keep[3]), C.byref(ep), 2, P(out), B, cin, H, W, co, co, stream())
    assert_close(from_nhwc(out), conv_ref(x, w, None, scale, shift, None, temb, 2), 1e-6, "first conv")
    # final 1x1: NHWC (64 or 8 channels) -> NCHW
    for c in (8, 64, 72):
        xin, wf, bf = rnd((B, c, H, W), 6), rnd((cin, c), 7, c ** -0.5), rnd((cin,), 8)
        o = torch.empty((B, cin, H, W), device="cuda")
        call("ddpm_conv1x1_final_fwd", P(to_nhwc(xin)), c, P(wf.cuda()), P(bf.cuda()), P(o), B, H, W, c, cin, stream())
        assert_close(o.cpu(), F.conv2d(xin, wf[:, :, None, None], bf), 1e-6, "final conv")


@pytest.mark.parametrize("case", [(2, 8, 8, 16), (3, 7, 7, 8), (2, 15, 9, 64), (1, 28, 28, 128)])
def test_pool_skip_and_add_temb(case):
    B, H, W, c = case
    x, temb = rnd((B, c, H, W), 1), rnd((B, c), 2)
    skip = torch.full((B, H, W, c + 4), 9.0, device="cuda")
    pooled = torch.empty((B, H // 2, W // 2, c), device="cuda")
    amax = torch.empty((B, H // 2, W // 2, c), dtype=torch.uint8, device="cuda")
    xd, td = to_nhwc(x), temb.cuda()
    call("ddpm_pool_skip_fwd", P(xd), c, P(td), P(skip), c + 4, P(pooled), c, P(amax), B, H, W, c, stream())
    assert torch.equal(from_nhwc(skip[..., :c]), x + temb[:, :, None, None])
    want, idx = F.max_pool2d(x, 2, 2, return_indices=True)
    assert torch.equal(from_nhwc(pooled), want)
    iy, ix = idx // W, idx % W
    assert torch.equal(from_nhwc(amax).long(), (iy % 2) * 2 + (ix % 2))
    out = torch.empty((B, H, W, c), device="cuda")
    call("ddpm_add_temb", P(xd), c, P(td), P(out), c, B, H * W, c, stream())
    assert torch.equal(from_nhwc(out), x + temb[:, :, None, None])


# ---------------------------------------------------------------- attention
@pytest.mark.parametrize("case", [(2, 49, 64, 8), (2, 196, 32, 4), (1, 1024, 128, 16), (2, 64, 512, 64), (2, 100, 8, 1),
                                  (1, 784, 128, 16), (3, 33, 256, 32),
                                  # N = 4096 (64x64 feature maps): the split-bf16 kernel at its longest sequence
                                  (1, 4096, 64, 8), (1, 4096, 128, 16),
                                  # edges of the software-pipelined kernel (C >= 128, N >= 128): one, two and three key tiles, a
                                  # ragged last tile, a last channel block that is partly empty, the widest query depth it takes
                                  (2, 128, 128, 16), (3, 129, 128, 16), (2, 160, 256, 32), (2, 161, 192, 32), (1, 200, 160, 8), (2, 95 + 33, 320, 32),
                                  # 128 query channels: the bottleneck of a base_channels = 128 network (C = 1024)
                                  (2, 64, 1024, 128), (2, 49, 1024, 128), (1, 256, 1024, 128), (3, 16, 1024, 128),
                                  # small maps: one workgroup per sample (forward up to 32 pixels, backward up to 64)
                                  (5, 9, 512, 64), (3, 16, 512, 64), (2, 32, 512, 64), (4, 1, 64, 8), (3, 25, 32, 4), (2, 32, 8, 1), (3, 49, 512, 64)])
def test_attention_fwd(case):
    B, N, c, dq = case
    dqp = (dq + 7) // 8 * 8
    x = rnd((B, N, c), 1)
    q, k, v = rnd((B, N, dq), 2), rnd((B, N, dq), 3), rnd((B, N, c), 4)
    gamma, temb = torch.tensor([0.37]), rnd((B, c), 5)
    qkv = torch.zeros((B, N, 2 * dqp + c))
    qkv[..., :dq], qkv[..., dqp:dqp + dq], qkv[..., 2 * dqp:] = q, k, v
    a = torch.softmax(torch.bmm(q.double(), k.double().transpose(1, 2)), dim=-1)
    raw_want = torch.bmm(a, v.double())
    want = gamma.double() * raw_want + x.double() + temb.double()[:, None, :]
    lse_want = torch.logsumexp(torch.bmm(q.double(), k.double().transpose(1, 2)), dim=-1)
    out = torch.empty((B, N, c), device="cuda")
    raw = torch.empty((B, N, c), device="cuda")
    lse = torch.empty((B, N), device="cuda")
    keep = [qkv.cuda(), x.cuda(), gamma.cuda(), temb.cuda()]
    call("ddpm_attention_fwd", P(keep[0]), P(keep[1]), c, P(keep[2]), P(out), c, P(raw), P(lse), P(keep[3]), B, N, c, dqp, stream())
    assert_close(out.cpu(), want, TOL, f"attention {case}")
    assert_close(raw.cpu(), raw_want, TOL, f"attention raw {case}")
    assert_close(lse.cpu(), lse_want, TOL, f"attention lse {case}")


@pytest.mark.parametrize("case", [(1, 1024, 128, 16), (1, 784, 128, 16), (1, 4096, 64, 8), (1, 4096, 128, 16), (2, 128, 128, 16), (3, 129, 128, 16),
                                  (1, 200, 160, 8), (2, 333, 64, 8), (2, 257, 96, 8), (2, 160, 256, 32)])
def test_attention_fwd_inference_call(case):
    """The call an eval-mode forward makes: no log-sum-exp, no raw output. With the fp16-pair arithmetic these calls (query
    depth up to 16) run the pair kernels, phase-by-phase and software-pipelined; same references and tolerance as above,
    and the gated output must agree with the training call's (three-plane bf16 kernels) to the same tolerance."""
    B, N, c, dq = case
    dqp = (dq + 7) // 8 * 8
    x = rnd((B, N, c), 11)
    q, k, v = rnd((B, N, dq), 12), rnd((B, N, dq), 13) * 2.0, rnd((B, N, c), 14)
    gamma, temb = torch.tensor([0.61]), rnd((B, c), 15)
    qkv = torch.zeros((B, N, 2 * dqp + c))
    qkv[..., :dq], qkv[..., dqp:dqp + dq], qkv[..., 2 * dqp:] = q, k, v
    a = torch.softmax(torch.bmm(q.double(), k.double().transpose(1, 2)), dim=-1)
    want = gamma.double() * torch.bmm(a, v.double()) + x.double() + temb.double()[:, None, :]
    keep = [qkv.cuda(), x.cuda(), gamma.cuda(), temb.cuda()]
    out = torch.full((B, N, c), float("nan"), device="cuda")
    call("ddpm_attention_fwd", P(keep[0]), P(keep[1]), c, P(keep[2]), P(out), c, None, None, P(keep[3]), B, N, c, dqp, stream())
    assert_close(out.cpu(), want, TOL, f"attention, inference call {case}")
    out2, raw, lse = torch.empty_like(out), torch.empty_like(out), torch.empty((B, N), device="cuda")
    call("ddpm_attention_fwd", P(keep[0]), P(keep[1]), c, P(keep[2]), P(out2), c, P(raw), P(lse), P(keep[3]), B, N, c, dqp, stream())
    assert_close(out.cpu(), out2.cpu().double(), TOL, f"attention, inference against training call {case}")


@pytest.mark.parametrize("case", [(16, 1024, 128, True), (64, 256, 128, False), (16, 1024, 64, True), (66, 256, 64, False),
                                  # shapes that the entry hands to the two plain calls: few rows, N not a multiple of 32, 256 channels
                                  (2, 256, 128, True), (24, 784, 128, True), (64, 256, 256, False)])
def test_self_attention_infer_is_projection_plus_attention(case):
    """ddpm_self_attention_infer against ddpm_conv1x1_fwd + ddpm_attention_fwd on the same packed weights: the same BITS,
    whether it hands k and v over as fp16 pairs (the first four shapes in the default arithmetic) or runs the two calls
    itself; and against the float64 attention of the float64 projection to the usual tolerance."""
    B, N, c, with_temb = case
    dq = c // 8
    dqp = (dq + 7) // 8 * 8
    lq = 2 * dqp + c
    x = rnd((B, N, c), 31)
    qw, kw, vw = rnd((dq, c), 32, c ** -0.5), rnd((dq, c), 33, 2.0 * c ** -0.5), rnd((c, c), 34, c ** -0.5)
    qb, kb, vb = rnd((dq,), 35, 0.1), rnd((dq,), 36, 0.1), rnd((c,), 37, 0.1)
    gamma, temb = torch.tensor([0.43]), rnd((B, c), 38)
    keep = [t.cuda() for t in (qw, kw, vw, qb, kb, vb, x, gamma, temb)]
    wf = torch.empty(packed_floats(1, lq, c), device="cuda")
    wt = torch.empty(packed_floats(1, c, lq), device="cuda")
    bias = torch.empty(lq, device="cuda")
    call("ddpm_pack_qkv", *(P(t) for t in keep[:6]), c, dq, dqp, P(wf), P(wt), P(bias), stream())
    tp = P(keep[8]) if with_temb else None
    # the two calls
    qkv = torch.empty((B, N, lq), device="cuda")
    d = ConvDesc(batch=B, in_h=N, in_w=1, c1=c, ld1=c, c2=0, ld2=0, cout=lq, ldo=lq, act=0, ldr=0, out_h=N, out_w=1, off_y=0, off_x=0)
    ep = Epilogue(None, P(bias), None, None)
    call("ddpm_conv1x1_fwd", C.byref(d), P(keep[6]), None, P(wf), C.byref(ep), P(qkv), stream())
    two = torch.full((B, N, c), float("nan"), device="cuda")
    call("ddpm_attention_fwd", P(qkv), P(keep[6]), c, P(keep[7]), P(two), c, None, None, tp, B, N, c, dqp, stream())
    # the one entry
    ws = torch.full((B, N, lq), float("nan"), device="cuda")
    one = torch.full((B, N, c), float("nan"), device="cuda")
    call("ddpm_self_attention_infer", P(keep[6]), c, P(wf), P(bias), P(keep[7]), tp, P(one), c, P(ws), B, N, c, dqp, stream())
    torch.cuda.synchronize()
    assert torch.equal(one, two), f"{case}: {(one - two).abs().max().item():.3e} apart"
    xd = x.double()
    q, k, v = xd @ qw.double().T + qb.double(), xd @ kw.double().T + kb.double(), xd @ vw.double().T + vb.double()
    want = gamma.double() * torch.bmm(torch.softmax(torch.bmm(q, k.transpose(1, 2)), dim=-1), v) + xd
    if with_temb:
        want = want + temb.double()[:, None, :]
    assert_close(one.cpu(), want, TOL, f"self-attention block {case}")


def test_attention_large_logits_rescale():
    """One key dominates from a late tile on: forces the online-softmax rescale branch."""
    B, N, c, dq = 1, 160, 32, 8
    q, k, v, x = rnd((B, N, dq), 1), rnd((B, N, dq), 2), rnd((B, N, c), 3), rnd((B, N, c), 4)
    k[0, 130] = q[0, 5] * 8.0          # key 130 (5th tile) spikes for query 5 (and correlates with others)
    k[0, 3] = q[0, 70] * 6.0
    qkv = torch.cat([q, k, v], -1)
    a = torch.softmax(torch.bmm(q.double(), k.double().transpose(1, 2)), dim=-1)
    want = 1.0 * torch.bmm(a, v.double()) + x.double()
    out = torch.empty((B, N, c), device="cuda")
    keep = [qkv.cuda(), x.cuda(), torch.ones(1, device="cuda")]
    call("ddpm_attention_fwd", P(keep[0]), P(keep[1]), c, P(keep[2]), P(out), c, None, None, None, B, N, c, dq, stream())
    assert_close(out.cpu(), want, TOL, "attention rescale")


@pytest.mark.parametrize("case", [(1, 1024, 128, 16), (1, 1024, 64, 8), (2, 300, 256, 32), (1, 784, 128, 16)])
def test_attention_scores_that_rise_from_tile_to_tile(case):
    """The matrix kernels let their running row maximum lag behind the scores by up to 5.5 before they rescale the output
    accumulators. Here every query's scores rise by 0.5 ... 1.5 per tile of 32 keys (up to 48 at the last key), so that the
    maximum lags, catches up every few tiles, and does so at different tiles for different queries; output, raw output
    and log-sum-exp against float64, inference and training call."""
    B, N, c, dq = case
    dqp = (dq + 7) // 8 * 8
    g = torch.Generator().manual_seed(21)
    u = torch.randn(dq, generator=g)
    u /= u.norm()
    amp = 0.5 + torch.rand(B, N, 1, generator=g)                                    # per query: 0.5 ... 1.5 per tile
    q = u * amp + 0.05 * torch.randn(B, N, dq, generator=g)
    k = u * (torch.arange(N, dtype=torch.float32) / 32.0)[None, :, None] + 0.05 * torch.randn(B, N, dq, generator=g)
    v, x = rnd((B, N, c), 3), rnd((B, N, c), 4)
    qkv = torch.zeros((B, N, 2 * dqp + c))
    qkv[..., :dq], qkv[..., dqp:dqp + dq], qkv[..., 2 * dqp:] = q, k, v
    scores = torch.bmm(q.double(), k.double().transpose(1, 2))
    raw_want = torch.bmm(torch.softmax(scores, dim=-1), v.double())
    want = 0.8 * raw_want + x.double()
    lse_want = torch.logsumexp(scores, dim=-1)
    keep = [qkv.cuda(), x.cuda(), torch.full((1,), 0.8, device="cuda")]
    out = torch.full((B, N, c), float("nan"), device="cuda")
    call("ddpm_attention_fwd", P(keep[0]), P(keep[1]), c, P(keep[2]), P(out), c, None, None, None, B, N, c, dqp, stream())
    assert_close(out.cpu(), want, TOL, f"rising scores, inference call {case}")
    out2, raw, lse = torch.empty_like(out), torch.empty_like(out), torch.empty((B, N), device="cuda")
    call("ddpm_attention_fwd", P(keep[0]), P(keep[1]), c, P(keep[2]), P(out2), c, P(raw), P(lse), None, B, N, c, dqp, stream())
    assert_close(out2.cpu(), want, TOL, f"rising scores, training call {case}")
    assert_close(raw.cpu(), raw_want, TOL, f"rising scores, raw output {case}")
    assert_close(lse.cpu(), lse_want, TOL, f"rising scores, log-sum-exp {case}")


def test_adam_step_argument_checks():
    """ddpm_adam_step through the raw C ABI: an empty list is a no-op, bad arguments return an error code."""
    from diffusion_model_amd import _lib
    h = _lib.lib()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    assert h.ddpm_adam_step(0, None, None, None, None, None, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, s) == 0
    init = [1.0, 1.0, 0.0, 0.0]                                                          # param, grad, exp_avg, exp_avg_sq
    t = [torch.full((5,), v, device="cuda") for v in init]
    arrs = [(C.c_void_p * 1)(x.data_ptr()) for x in t]
    n = (C.c_int64 * 1)(5)
    assert h.ddpm_adam_step(1, *arrs, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 0, s) != 0          # step counts from 1
    assert h.ddpm_adam_step(1, *arrs, n, 1e-3, 1.0, 0.999, 1e-8, 0.0, 1, s) != 0          # beta1 < 1
    assert h.ddpm_adam_step(1, arrs[0], None, arrs[2], arrs[3], n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, s) != 0
    assert b"invalid argument" in h.ddpm_last_error()
    torch.cuda.synchronize()
    assert all(torch.equal(x, torch.full((5,), v, device="cuda")) for x, v in zip(t, init))   # nothing was touched
    assert h.ddpm_adam_step(1, *arrs, n, 1e-3, 0.9, 0.999, 1e-8, 0.0, 1, s) == 0
    torch.cuda.synchronize()
    # first step of Adam from m = v = 0 with g = 1: the update is lr * 1 / (1 + eps)
    torch.testing.assert_close(t[0], torch.full((5,), 1.0 - 1e-3, device="cuda"), rtol=1e-6, atol=0)
