import sys, os, ctypes as C, torch
import torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from gpu_util import ConvDesc, Epilogue, P, call, from_nhwc, packed_floats, rnd, stream, to_nhwc
torch.manual_seed(0)
for (B, H, W, ci, co, scale, use_res) in [(4, 8, 8, 256, 512, 1e-5, False), (4, 8, 8, 256, 512, 1.0, False), (4, 16, 16, 128, 256, 1.0, False), (4, 32, 32, 64, 128, 1.0, False), (4, 32, 32, 64, 64, 1.0, False), (4, 32, 32, 64, 64, 1e-5, False), (4, 32, 32, 128, 64, 1e-5, True),
                                          (4, 16, 16, 128, 128, 1e-5, True), (4, 16, 16, 256, 128, 1e-5, False), (4, 8, 8, 256, 256, 1e-5, True),
                                          (4, 8, 8, 512, 256, 1e-5, False), (4, 8, 8, 128, 256, 1e-5, False)]:
    w = torch.randn(co, ci, 3, 3) * (9 * ci) ** -0.5
    dy = torch.randn(B, co, H, W) * scale
    res = torch.randn(B, ci, H, W) * scale if use_res else None
    want = F.conv_transpose2d(dy.double(), w.double(), padding=1)
    if use_res:
        want = want + res.double()
    rows, depth = ci, co
    wp = torch.empty(packed_floats(9, rows, depth), device="cuda")
    call("ddpm_pack_conv3x3", P(w.cuda().contiguous()), co, ci, 1, P(wp), stream())
    dyd = to_nhwc(dy)
    dx = torch.empty((B, H, W, ci), device="cuda")
    resd = None
    ldr = 0
    if use_res:   # strided residual: left half of a wider buffer
        buf = torch.zeros((B, H, W, 2 * ci), device="cuda")
        buf[..., :ci] = to_nhwc(res)
        resd, ldr = buf, 2 * ci
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=co, ld1=co, c2=0, ld2=0, cout=ci, ldo=ci, act=0, ldr=ldr, out_h=H, out_w=W)
    ep = Epilogue(None, None, P(resd), None)
    call("ddpm_conv3x3_fwd", C.byref(d), P(dyd), None, P(wp), C.byref(ep), P(dx), stream())
    got = from_nhwc(dx).double()
    err = (got - want).abs().max().item() / want.abs().max().item()
    print(f"B={B} {H}x{W} dy{co}->dx{ci} scale={scale} res={use_res}: rel err {err:.2e}")
