"""Backward kernels on the GPU, through the C ABI, against torch autograd on the CPU (fp32/fp64).

Tolerance: 3e-5 of the tensor's scale for MFMA contractions (weight gradients sum over up to
~10^5 pixels in a different order than the CPU), 1e-5 for the float64-accumulated reductions.
"""
import ctypes as C

import pytest
import torch
import torch.nn.functional as F

from diffusion_model_amd import _lib
from diffusion_model_amd._lib import EmbedBwdArgs
from gpu_util import ConvDesc, Epilogue, P, assert_close, call, from_nhwc, packed_floats, rnd, stream, to_nhwc

pytestmark = pytest.mark.gpu
TOL = 3e-5


def workspace(c):
    n = _lib.lib().ddpm_reduce_workspace_doubles(c)
    return torch.empty(n, dtype=torch.float64, device="cuda")


@pytest.mark.parametrize("case", [(2, 9, 7, 8, 0, False, False), (3, 28, 28, 64, 1, False, True), (2, 14, 14, 128, 2, True, True),
                                  (4, 7, 7, 512, 2, False, False), (1, 5, 5, 16, 2, True, False)])
def test_batchnorm_train_fwd_bwd(case):
    B, H, W, c, act, use_res, use_temb = case
    x = (rnd((B, c, H, W), 1) * 1.5 + 0.3).requires_grad_(True)
    w, b = (rnd((c,), 2).abs() + 0.5).requires_grad_(True), rnd((c,), 3).requires_grad_(True)
    res = rnd((B, c, H, W), 4).requires_grad_(True) if use_res else None
    temb = rnd((B, c), 5) if use_temb else None
    rm, rv = rnd((c,), 6) * 0.1, rnd((c,), 7).abs() + 0.5
    rm_ref, rv_ref = rm.clone(), rv.clone()
    z = F.batch_norm(x, rm_ref, rv_ref, w, b, training=True, momentum=0.1, eps=1e-5)
    if use_res:
        z = z + res
    y = F.relu(z) if act == 1 else (F.gelu(z) if act == 2 else z)
    if use_temb:
        y = y + temb[:, :, None, None]
    dout = rnd((B, c, H, W), 8)
    y.backward(dout)

    xd = to_nhwc(x.detach())
    ws = workspace(c)
    mean, invstd = torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    rmd, rvd = rm.cuda(), rv.cuda()
    rows = B * H * W
    call("ddpm_bn_stats", P(xd), c, rows, c, 1e-5, 0.1, P(ws), P(mean), P(invstd), P(rmd), P(rvd), stream())
    assert_close(rmd.cpu(), rm_ref, 1e-6, "running_mean")
    assert_close(rvd.cpu(), rv_ref, 1e-6, "running_var")
    out = torch.empty((B, H, W, c), device="cuda")
    wd, bd = w.detach().cuda(), b.detach().cuda()
    resd = to_nhwc(res.detach()) if use_res else None
    call("ddpm_bn_apply_fwd", P(xd), c, P(mean), P(invstd), P(wd), P(bd), P(resd), c, P(temb.cuda() if use_temb else None), act,
         P(out), c, B, H * W, c, stream())
    assert_close(from_nhwc(out), y.detach(), 1e-5, "bn fwd")
    sums, dw, db = torch.empty(2 * c, device="cuda"), torch.empty(c, device="cuda"), torch.empty(c, device="cuda")
    dx = torch.empty((B, H, W, c), device="cuda")
    dres = torch.empty((B, H, W, c), device="cuda") if use_res else None
    dcb = torch.empty(c, device="cuda")
    call("ddpm_bn_bwd", P(xd), c, P(to_nhwc(dout)), c, P(resd), c, P(mean), P(invstd), P(wd), P(bd), act, rows, c, P(ws), P(sums),
         P(dw), P(db), 0, P(dx), c, P(dres), c, P(dcb), stream())
    assert_close(from_nhwc(dx), x.grad, 2e-5, "bn dx")
    assert_close(dw.cpu(), w.grad, 2e-5, "bn dw")
    assert_close(db.cpu(), b.grad, 2e-5, "bn db")
    if use_res:
        assert_close(from_nhwc(dres), res.grad, 2e-5, "bn dres")
    assert dcb.abs().max().item() <= 1e-3 * max(1.0, x.grad.abs().max().item()) * (rows ** 0.5)   # ~0 up to rounding


@pytest.mark.parametrize("case", [(4, 49, 16), (3, 784, 64), (2, 100, 512)])
def test_colsum_groups(case):
    B, hw, c = case
    x = rnd((B, hw, c), 1)
    ws = workspace(c)
    out = torch.zeros((B, c), device="cuda")
    xd = x.cuda()
    call("ddpm_colsum", P(xd), c, B, hw, c, P(ws), 0, P(out), stream())
    assert_close(out.cpu(), x.double().sum(1), 1e-6, "per-sample colsum")
    call("ddpm_colsum", P(xd), c, B, hw, c, P(ws), 1, P(out), stream())
    assert_close(out.cpu(), 2 * x.double().sum(1), 1e-6, "accumulating colsum")
    tot = torch.empty(c, device="cuda")
    call("ddpm_colsum", P(xd), c, 1, B * hw, c, P(ws), 0, P(tot), stream())
    assert_close(tot.cpu(), x.double().sum((0, 1)), 1e-6, "global colsum")


def pack3(w, flip):
    rows, depth = (w.shape[1], w.shape[0]) if flip else (w.shape[0], w.shape[1])
    out = torch.empty(packed_floats(9, rows, depth), device="cuda")
    call("ddpm_pack_conv3x3", P(w.cuda().contiguous()), w.shape[0], w.shape[1], flip, P(out), stream())
    return out


@pytest.mark.parametrize("case", [(2, 8, 8, 8, 16), (3, 28, 28, 64, 64), (2, 14, 14, 128, 256), (6, 7, 7, 512, 256), (2, 30, 30, 32, 8),
                                  (16, 28, 28, 64, 128), (6, 3, 3, 512, 512), (40, 1, 1, 128, 128), (3, 5, 3, 128, 96), (2, 2, 17, 72, 136),
                                  (5, 33, 1, 128, 64),
                                  # 64 channels (or fewer) on one side, 128 or more on the other: two taps per weight-gradient tile
                                  (4, 28, 28, 128, 64), (3, 9, 11, 64, 256), (3, 10, 7, 256, 64), (2, 5, 6, 48, 192), (2, 6, 5, 160, 40)])
def test_conv3x3_dgrad_wgrad(case):
    B, H, W, ci, co = case
    x = rnd((B, ci, H, W), 1).requires_grad_(True)
    w = rnd((co, ci, 3, 3), 2, (9 * ci) ** -0.5).requires_grad_(True)
    dy = rnd((B, co, H, W), 3)
    F.conv2d(x, w, padding=1).backward(dy)
    dyd, xd = to_nhwc(dy), to_nhwc(x.detach())
    # data gradient: a 3x3 conv of dy with flipped, role-swapped weights
    dx = torch.empty((B, H, W, ci), device="cuda")
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=co, ld1=co, c2=0, ld2=0, cout=ci, ldo=ci, act=0, ldr=0, out_h=H, out_w=W)
    ep = Epilogue(None, None, None, None)
    call("ddpm_conv3x3_fwd", C.byref(d), P(dyd), None, P(pack3(w.detach(), 1)), C.byref(ep), P(dx), stream())
    assert_close(from_nhwc(dx), x.grad, TOL, f"dgrad {case}")
    packed, gw = torch.empty(w.numel(), device="cuda"), torch.empty(w.shape, device="cuda")
    call("ddpm_conv3x3_wgrad", P(dyd), co, co, P(xd), ci, ci, B, H, W, P(packed), P(gw), stream())
    assert_close(gw.cpu(), w.grad, TOL, f"wgrad {case}")


@pytest.mark.parametrize("case", [(2, 49, 64, 80), (3, 784, 128, 160), (2, 64, 512, 640), (4, 100, 8, 24)])
def test_conv1x1_bwd(case):
    B, N, ci, co = case
    x = rnd((B, ci, N, 1), 1).requires_grad_(True)
    w = rnd((co, ci), 2, ci ** -0.5).requires_grad_(True)
    dy = rnd((B, co, N, 1), 3)
    F.conv2d(x, w[:, :, None, None]).backward(dy)
    dyd, xd = to_nhwc(dy), to_nhwc(x.detach())
    gw = torch.empty((co, ci), device="cuda")
    call("ddpm_conv1x1_wgrad", P(dyd), co, co, P(xd), ci, ci, B * N, P(gw), stream())
    assert_close(gw.cpu(), w.grad, TOL, f"1x1 wgrad {case}")
    dx = torch.empty((B, N, 1, ci), device="cuda")
    d = ConvDesc(batch=B, in_h=N, in_w=1, c1=co, ld1=co, c2=0, ld2=0, cout=ci, ldo=ci, act=0, ldr=0, out_h=N, out_w=1)
    ep = Epilogue(None, None, None, None)
    wt = torch.empty(packed_floats(1, ci, co), device="cuda")
    call("ddpm_pack_conv1x1", P(w.detach().cuda().contiguous()), ci, co, 1, P(wt), stream())
    call("ddpm_conv1x1_fwd", C.byref(d), P(dyd), None, P(wt), C.byref(ep), P(dx), stream())
    assert_close(from_nhwc(dx), x.grad, TOL, f"1x1 dgrad {case}")


@pytest.mark.parametrize("case", [(2, 7, 7, 64, 32, 14, 14), (2, 7, 7, 32, 16, 15, 15), (2, 16, 16, 32, 16, 16, 16),
                                  (3, 14, 14, 128, 64, 28, 28), (2, 7, 5, 16, 8, 17, 9), (2, 7, 7, 128, 128, 13, 13), (9, 1, 1, 256, 128, 3, 3),
                                  (4, 3, 3, 512, 256, 7, 7)])
def test_convT_bwd(case):
    B, H, W, ci, co, OH, OW = case
    x = rnd((B, ci, H, W), 1).requires_grad_(True)
    w = rnd((ci, co, 2, 2), 2, ci ** -0.5).requires_grad_(True)
    b = rnd((co,), 3).requires_grad_(True)
    y = F.conv_transpose2d(x, w, b, stride=2)
    dyp, dxp = OH - y.shape[2], OW - y.shape[3]
    yp = F.pad(y, [dxp // 2, dxp - dxp // 2, dyp // 2, dyp - dyp // 2])
    dout = rnd((B, co, OH, OW), 4)
    yp.backward(dout)
    dd, xd = to_nhwc(dout), to_nhwc(x.detach())
    wp = torch.empty(packed_floats(4, ci, co), device="cuda")
    call("ddpm_pack_convT2x2_dgrad", P(w.detach().cuda()), ci, co, P(wp), stream())
    dx = torch.empty((B, H, W, ci), device="cuda")
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=co, ld1=co, c2=0, ld2=0, cout=ci, ldo=ci, act=0, ldr=0, out_h=OH, out_w=OW,
                 off_y=dyp // 2, off_x=dxp // 2)
    ep = Epilogue(None, None, None, None)
    call("ddpm_convT2x2_dgrad", C.byref(d), P(dd), P(wp), C.byref(ep), P(dx), stream())
    assert_close(from_nhwc(dx), x.grad, TOL, f"convT dgrad {case}")
    packed, gw = torch.empty(w.numel(), device="cuda"), torch.empty(w.shape, device="cuda")
    call("ddpm_convT2x2_wgrad", P(dd), co, co, OH, OW, dyp // 2, dxp // 2, P(xd), ci, ci, B, H, W, P(packed), P(gw), stream())
    assert_close(gw.cpu(), w.grad, TOL, f"convT wgrad {case}")
    # bias gradient: column sum over the window of the output that the transposed conv wrote
    y0, x0 = max(dyp // 2, 0), max(dxp // 2, 0)
    y1, x1 = min(OH, dyp // 2 + 2 * H), min(OW, dxp // 2 + 2 * W)
    gb = torch.empty(co, device="cuda")
    call("ddpm_colsum_window", P(dd), co, B, OH, OW, y0, x0, y1 - y0, x1 - x0, co, P(workspace(co)), P(gb), stream())
    assert_close(gb.cpu(), b.grad, 1e-5, f"convT dbias {case}")


@pytest.mark.parametrize("cin", [1, 3])
def test_edge_convs_bwd(cin):
    B, H, W, c = 3, 12, 10, 64
    x = rnd((B, cin, H, W), 1)
    w = rnd((c, cin, 3, 3), 2, 0.3).requires_grad_(True)
    dy = rnd((B, c, H, W), 3)
    F.conv2d(x, w, padding=1).backward(dy)
    gw = torch.empty(w.shape, device="cuda")
    call("ddpm_conv3x3_first_wgrad", P(to_nhwc(dy)), c, P(x.cuda()), P(workspace(c)), P(gw), B, cin, H, W, c, stream())
    assert_close(gw.cpu(), w.grad, 1e-5, "first conv wgrad")
    # final 1x1
    xin = rnd((B, c, H, W), 4).requires_grad_(True)
    wf, bf = rnd((cin, c), 5, c ** -0.5).requires_grad_(True), rnd((cin,), 6).requires_grad_(True)
    dout = rnd((B, cin, H, W), 7)
    F.conv2d(xin, wf[:, :, None, None], bf).backward(dout)
    dx, dw, db = torch.empty((B, H, W, c), device="cuda"), torch.empty((cin, c), device="cuda"), torch.empty(cin, device="cuda")
    call("ddpm_conv1x1_final_bwd", P(dout.cuda()), P(to_nhwc(xin.detach())), c, P(wf.detach().cuda()), P(workspace(c)), P(dx), c, P(dw),
         P(db), B, H, W, c, cin, stream())
    assert_close(from_nhwc(dx), xin.grad, 1e-5, "final dx")
    assert_close(dw.cpu(), wf.grad, 1e-5, "final dw")
    assert_close(db.cpu(), bf.grad, 1e-5, "final db")


@pytest.mark.parametrize("case", [(2, 8, 8, 16), (3, 7, 7, 8), (2, 15, 9, 64)])
def test_pool_skip_bwd(case):
    B, H, W, c = case
    x = rnd((B, c, H, W), 1).requires_grad_(True)
    dskip, dpool = rnd((B, c, H, W), 2), rnd((B, c, H // 2, W // 2), 3)
    (x * 1.0).backward(dskip)
    F.max_pool2d(x, 2, 2).backward(dpool)
    xd = to_nhwc(x.detach())
    pooled = torch.empty((B, H // 2, W // 2, c), device="cuda")
    amax = torch.empty((B, H // 2, W // 2, c), dtype=torch.uint8, device="cuda")
    call("ddpm_pool_skip_fwd", P(xd), c, None, None, 0, P(pooled), c, P(amax), B, H, W, c, stream())
    dx = torch.empty((B, H, W, c), device="cuda")
    call("ddpm_pool_skip_bwd", P(to_nhwc(dskip)), c, P(to_nhwc(dpool)), c, P(amax), P(dx), c, B, H, W, c, stream())
    assert torch.equal(from_nhwc(dx), x.grad)


@pytest.mark.parametrize("n_labels", [0, 1, 5])
def test_time_embed_fwd_bwd(n_labels):
    B, E, chans, ncls = 5, 16, [8, 16, 32, 64], 10
    t = torch.tensor([3, 999, 0, 500, 17])
    from oracle import ddpm_oracle as O
    emb = O.time_embedding(t, E)
    Ws = [rnd((c, E), 10 + k, 0.3).requires_grad_(True) for k, c in enumerate(chans)]
    bs = [rnd((c,), 20 + k).requires_grad_(True) for k, c in enumerate(chans)]
    LWs = [rnd((c, E), 30 + k, 0.3).requires_grad_(True) for k, c in enumerate(chans)]
    Lbs = [rnd((c,), 40 + k).requires_grad_(True) for k, c in enumerate(chans)]
    table = rnd((ncls, E), 50).requires_grad_(True)
    y = torch.tensor([3, 7, 3, 0, 9])[:n_labels] if n_labels else None
    outs = []
    for k in range(4):
        o = F.linear(emb, Ws[k], bs[k])
        if n_labels:
            o = o + F.linear(F.embedding(y, table), LWs[k], Lbs[k])
        outs.append(o)
    dts = [rnd((B, c), 60 + k) for k, c in enumerate(chans)]
    sum((o * d).sum() for o, d in zip(outs, dts)).backward()

    from diffusion_model_amd._lib import EmbedArgs
    a = EmbedArgs()
    a.batch, a.emb_dim, a.n_labels, a.t_is_float, a.n_classes = B, E, n_labels, 0, ncls
    keep = [t.cuda(), y.cuda() if n_labels else None, table.detach().cuda()]
    a.t = keep[0].data_ptr()
    if n_labels:
        a.y, a.label_table = keep[1].data_ptr(), keep[2].data_ptr()
    dev = lambda lst: [v.detach().cuda() for v in lst]
    Wd, bd, LWd, Lbd = dev(Ws), dev(bs), dev(LWs), dev(Lbs)
    od = [torch.empty((B, c), device="cuda") for c in chans]
    embd = torch.empty((B, E), device="cuda")
    for k in range(4):
        a.ch[k], a.w[k], a.b[k], a.out[k] = chans[k], Wd[k].data_ptr(), bd[k].data_ptr(), od[k].data_ptr()
        if n_labels:
            a.lw[k], a.lb[k] = LWd[k].data_ptr(), Lbd[k].data_ptr()
    a.emb_out = embd.data_ptr()
    call("ddpm_time_embed", C.byref(a), stream())
    for k in range(4):
        assert_close(od[k].cpu(), outs[k].detach(), 2e-6, f"temb{k}")
    assert_close(embd.cpu(), emb, 2e-6, "emb")

    g = EmbedBwdArgs()
    g.batch, g.emb_dim, g.n_labels, g.n_classes = B, E, n_labels, ncls
    g.emb = embd.data_ptr()
    dtd = dev(dts)
    dW, dB = [torch.empty_like(v) for v in Wd], [torch.empty_like(v) for v in bd]
    dLW, dLB = [torch.empty_like(v) for v in LWd], [torch.empty_like(v) for v in Lbd]
    dtab = torch.zeros((ncls, E), device="cuda")
    if n_labels:
        g.y, g.label_table, g.dlabel_table = keep[1].data_ptr(), keep[2].data_ptr(), dtab.data_ptr()
    for k in range(4):
        g.ch[k], g.dtemb[k], g.dw[k], g.db[k] = chans[k], dtd[k].data_ptr(), dW[k].data_ptr(), dB[k].data_ptr()
        if n_labels:
            g.lw[k], g.dlw[k], g.dlb[k] = LWd[k].data_ptr(), dLW[k].data_ptr(), dLB[k].data_ptr()
    call("ddpm_time_embed_bwd", C.byref(g), stream())
    for k in range(4):
        assert_close(dW[k].cpu(), Ws[k].grad, 1e-5, f"dW{k}")
        assert_close(dB[k].cpu(), bs[k].grad, 1e-5, f"db{k}")
        if n_labels:
            assert_close(dLW[k].cpu(), LWs[k].grad, 1e-5, f"dLW{k}")
            assert_close(dLB[k].cpu(), Lbs[k].grad, 1e-5, f"dLb{k}")
    if n_labels:
        assert_close(dtab.cpu(), table.grad, 1e-5, "d label table")


@pytest.mark.parametrize("case", [(2, 49, 64, 8), (2, 196, 32, 4), (1, 784, 128, 16), (2, 64, 512, 64), (2, 100, 8, 1), (3, 33, 256, 32), (2, 196, 256, 32),
                                  (1, 1024, 128, 16), (1, 4096, 64, 8), (1, 4096, 128, 16),
                                  (2, 64, 1024, 128), (2, 49, 1024, 128), (1, 256, 1024, 128), (3, 16, 1024, 128),
                                  # small maps: one workgroup per sample (forward up to 32 pixels, backward up to 64)
                                  (5, 9, 512, 64), (3, 16, 512, 64), (2, 32, 512, 64), (4, 1, 64, 8), (3, 25, 32, 4), (2, 32, 8, 1), (3, 49, 512, 64)])
def test_attention_bwd(case):
    B, N, c, dq = case
    dqp = (dq + 7) // 8 * 8
    x = rnd((B, N, c), 1)
    q, k, v = (rnd((B, N, dq), 2) * 0.7).requires_grad_(True), (rnd((B, N, dq), 3) * 0.7).requires_grad_(True), rnd((B, N, c), 4).requires_grad_(True)
    gamma = torch.tensor([0.37], dtype=torch.float64, requires_grad=True)
    a = torch.softmax(torch.bmm(q.double(), k.double().transpose(1, 2)), dim=-1)
    y = gamma * torch.bmm(a, v.double()) + x.double()
    dy = rnd((B, N, c), 5)
    y.backward(dy.double())

    qkv = torch.zeros((B, N, 2 * dqp + c))
    qkv[..., :dq], qkv[..., dqp:dqp + dq], qkv[..., 2 * dqp:] = q.detach(), k.detach(), v.detach()
    out, raw, lse = (torch.empty((B, N, c), device="cuda"), torch.empty((B, N, c), device="cuda"), torch.empty((B, N), device="cuda"))
    qd, xd, gd = qkv.cuda(), x.cuda(), gamma.detach().float().cuda()
    call("ddpm_attention_fwd", P(qd), P(xd), c, P(gd), P(out), c, P(raw), P(lse), None, B, N, c, dqp, stream())
    dqkv = torch.full((B, N, 2 * dqp + c), float("nan"), device="cuda")
    dgamma = torch.empty(1, device="cuda")
    ws, es = torch.empty(4096, dtype=torch.float64, device="cuda"), torch.empty((B, N), device="cuda")
    call("ddpm_attention_bwd", P(qd), P(dy.cuda()), c, P(raw), P(lse), P(gd), P(ws), P(es), P(dqkv), P(dgamma), B, N, c, dqp, stream())
    got = dqkv.cpu()
    assert_close(got[..., :dq], q.grad, TOL, f"dq {case}")
    assert_close(got[..., dqp:dqp + dq], k.grad, TOL, f"dk {case}")
    assert_close(got[..., 2 * dqp:], v.grad, TOL, f"dv {case}")
    assert torch.all(got[..., dq:dqp] == 0) and torch.all(got[..., dqp + dq:2 * dqp] == 0)
    assert abs(dgamma.item() - gamma.grad.item()) <= 1e-4 * max(1.0, abs(gamma.grad.item()))


@pytest.mark.parametrize("case", [(2, 200, 64, 8), (1, 170, 128, 16), (2, 136, 256, 32)])
def test_attention_bwd_very_negative_scores_ragged(case):
    """Every score is around -150 and the sequence does not fill its last 32-row tile: exp(0 - lse) of a key row past
    the end overflows to infinity. The kernels do not mask such rows where they only reach their own (unstored)
    output column, and must mask them where they meet real data; no NaN may come out."""
    B, N, c, dq = case
    base = torch.ones(dq) * (150.0 / dq) ** 0.5
    q = (base + 0.05 * rnd((B, N, dq), 2)).requires_grad_(True)
    k = (-base + 0.05 * rnd((B, N, dq), 3)).requires_grad_(True)
    v = rnd((B, N, c), 4).requires_grad_(True)
    x = rnd((B, N, c), 1)
    gamma = torch.tensor([0.8], dtype=torch.float64, requires_grad=True)
    sc = torch.bmm(q.double(), k.double().transpose(1, 2))
    assert sc.max() < -120
    y = gamma * torch.bmm(torch.softmax(sc, dim=-1), v.double()) + x.double()
    dy = rnd((B, N, c), 5)
    y.backward(dy.double())
    qkv = torch.cat([q.detach(), k.detach(), v.detach()], -1)
    out, raw, lse = (torch.empty((B, N, c), device="cuda"), torch.empty((B, N, c), device="cuda"), torch.empty((B, N), device="cuda"))
    qd, xd, gd = qkv.cuda(), x.cuda(), gamma.detach().float().cuda()
    call("ddpm_attention_fwd", P(qd), P(xd), c, P(gd), P(out), c, P(raw), P(lse), None, B, N, c, dq, stream())
    assert_close(out.cpu(), y.detach(), TOL, f"forward {case}")
    dqkv = torch.full((B, N, 2 * dq + c), float("nan"), device="cuda")
    dgamma = torch.empty(1, device="cuda")
    ws, es = torch.empty(4096, dtype=torch.float64, device="cuda"), torch.empty((B, N), device="cuda")
    call("ddpm_attention_bwd", P(qd), P(dy.cuda()), c, P(raw), P(lse), P(gd), P(ws), P(es), P(dqkv), P(dgamma), B, N, c, dq, stream())
    got = dqkv.cpu()
    assert torch.isfinite(got).all()
    assert_close(got[..., :dq], q.grad, TOL, f"dq {case}")
    assert_close(got[..., dq:2 * dq], k.grad, TOL, f"dk {case}")
    assert_close(got[..., 2 * dq:], v.grad, TOL, f"dv {case}")


def test_attention_batch_beyond_the_grid_limit():
    """The sample index is the launch grid's z axis, which holds 65535: a larger batch (the reference's classes take any
    batch, second-iteration/unet.py:52-66) runs as consecutive launches and must give, for the samples on either side of
    the cut, the bits the same samples give alone."""
    B, N, c, dq = 65535 + 9, 64, 64, 8
    g = torch.Generator(device="cuda").manual_seed(3)
    qkv = torch.randn((B, N, 2 * dq + c), device="cuda", generator=g) * 0.3
    x = torch.randn((B, N, c), device="cuda", generator=g)
    dy = torch.randn((B, N, c), device="cuda", generator=g)
    gd, temb = torch.tensor([0.6], device="cuda"), torch.randn((B, c), device="cuda", generator=g)

    def run(lo, hi):
        b = hi - lo
        q, xx, dd, te = qkv[lo:hi].contiguous(), x[lo:hi].contiguous(), dy[lo:hi].contiguous(), temb[lo:hi].contiguous()
        out, raw, lse = torch.empty((b, N, c), device="cuda"), torch.empty((b, N, c), device="cuda"), torch.empty((b, N), device="cuda")
        call("ddpm_attention_fwd", P(q), P(xx), c, P(gd), P(out), c, P(raw), P(lse), P(te), b, N, c, dq, stream())
        dqkv = torch.full((b, N, 2 * dq + c), float("nan"), device="cuda")
        dgamma = torch.empty(1, device="cuda")
        ws, es = torch.empty(4096, dtype=torch.float64, device="cuda"), torch.empty((b, N), device="cuda")
        call("ddpm_attention_bwd", P(q), P(dd), c, P(raw), P(lse), P(gd), P(ws), P(es), P(dqkv), P(dgamma), b, N, c, dq, stream())
        torch.cuda.synchronize()
        return out, lse, dqkv

    out, lse, dqkv = run(0, B)
    assert torch.isfinite(out).all() and torch.isfinite(dqkv).all()
    lo = 65535 - 6
    o2, l2, d2 = run(lo, B)                      # the 15 samples around the cut, as a batch of their own
    assert torch.equal(out[lo:], o2) and torch.equal(lse[lo:], l2) and torch.equal(dqkv[lo:], d2)
