#!/usr/bin/env python3
"""Kernel micro-benchmarks through the C ABI (no model around them), for profiling runs.

    python tools/microbench.py conv3x3 B H W Cin Cout [iters]
    python tools/microbench.py wgrad   B H W Cin Cout [iters]
    python tools/microbench.py conv1x1 B H W Cin Cout [iters]
    python tools/microbench.py convT   B H W Cin Cout [iters]  (2x2 stride-2 transposed conv into the right half of a [B,2H,2W,2*Cout] buffer)
    python tools/microbench.py attn    B N C [iters]         (forward + backward)
    python tools/microbench.py attnblock B N C [iters]       (inference attention block: q|k|v projection + attention, one entry)
    python tools/microbench.py step    NELEMS [iters]        (reverse-step kernel, HBM-bound)
Prints the average launch time from HIP events and the achieved TFLOP/s or GB/s.
"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from diffusion_model_amd._lib import ConvDesc, Epilogue, call, packed_floats  # noqa: E402


def P(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def S():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timeit(fn, iters):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def main():
    kind = sys.argv[1]
    a = [int(v) for v in sys.argv[2:]]
    dev = "cuda"
    torch.manual_seed(0)     # (the digests some modes print compare builds of the library on the same inputs)
    if kind in ("conv3x3", "wgrad"):
        B, H, W, ci, co = a[:5]
        iters = a[5] if len(a) > 5 else 20
        x = torch.randn(B, H, W, ci, device=dev) * float(os.environ.get("MB_SCALE", "1"))   # (MB_SCALE=1e-4: operands whose fp16 part is subnormal)
        w = torch.randn(co, ci, 3, 3, device=dev) * (9 * ci) ** -0.5
        wp = torch.empty(packed_floats(9, co, ci), device=dev)
        call("ddpm_pack_conv3x3", P(w), co, ci, 0, P(wp), S())
        y = torch.empty(B, H, W, co, device=dev)
        flops = 2.0 * B * H * W * co * ci * 9
        if kind == "conv3x3":
            d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=int(os.environ.get("MB_ACT", "2")), ldr=0, out_h=H, out_w=W)
            sc, sh = torch.ones(co, device=dev), torch.zeros(co, device=dev)
            ep = Epilogue(P(sc), P(sh), None, None)
            t = timeit(lambda: call("ddpm_conv3x3_fwd", C.byref(d), P(x), None, P(wp), C.byref(ep), P(y), S()), iters)
        else:
            dy = torch.randn(B, H, W, co, device=dev)
            packed, gw = torch.empty(w.numel(), device=dev), torch.empty_like(w)
            t = timeit(lambda: call("ddpm_conv3x3_wgrad", P(dy), co, co, P(x), ci, ci, B, H, W, P(packed), P(gw), S()), iters)
        import hashlib
        digest = hashlib.sha1((y if kind == "conv3x3" else gw).cpu().numpy().tobytes()).hexdigest()[:12] if B * H * W * co <= (1 << 27) else "-"
        print(f"{kind} B={B} {H}x{W} {ci}->{co}: {t*1e6:.1f} us  {flops/t/1e12:.1f} TFLOP/s  {digest}")
    elif kind == "conv1x1":
        B, H, W, ci, co = a[:5]
        iters = a[5] if len(a) > 5 else 20
        x = torch.randn(B, H, W, ci, device=dev)
        w = torch.randn(co, ci, device=dev) * ci ** -0.5
        wp = torch.empty(packed_floats(1, co, ci), device=dev)
        call("ddpm_pack_conv1x1", P(w), co, ci, 0, P(wp), S())
        y = torch.empty(B, H, W, co, device=dev)
        d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=H, out_w=W)
        sh = torch.zeros(co, device=dev)
        ep = Epilogue(None, P(sh), None, None)
        t = timeit(lambda: call("ddpm_conv1x1_fwd", C.byref(d), P(x), None, P(wp), C.byref(ep), P(y), S()), iters)
        gb = 4.0 * B * H * W * (ci + co)
        print(f"conv1x1 B={B} {H}x{W} {ci}->{co}: {t*1e6:.1f} us  {2.0*B*H*W*ci*co/t/1e12:.1f} TFLOP/s  {gb/t/1e12:.2f} TB/s algorithmic")
    elif kind == "convT":
        B, H, W, ci, co = a[:5]
        iters = a[5] if len(a) > 5 else 20
        x = torch.randn(B, H, W, ci, device=dev)
        w = torch.randn(ci, co, 2, 2, device=dev) * ci ** -0.5
        wp = torch.empty(packed_floats(1, 4 * co, ci), device=dev)
        call("ddpm_pack_convT2x2", P(w), ci, co, P(wp), S())
        cat = torch.zeros(B, 2 * H, 2 * W, 2 * co, device=dev)
        right = cat[..., co:]
        d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=2 * co, act=0, ldr=0, out_h=2 * H, out_w=2 * W,
                     off_y=0, off_x=0)
        sh = torch.zeros(co, device=dev)
        ep = Epilogue(None, P(sh), None, None)
        t = timeit(lambda: call("ddpm_convT2x2_fwd", C.byref(d), P(x), P(wp), C.byref(ep), C.c_void_p(right.data_ptr()), S()), iters)
        gb = 4.0 * B * H * W * (ci + 4 * co)
        import hashlib
        digest = hashlib.sha1(cat.cpu().numpy().tobytes()).hexdigest()[:12] if B * H * W * co <= (1 << 26) else "-"
        print(f"convT B={B} {H}x{W} {ci}->{co}: {t*1e6:.1f} us  {2.0*B*H*W*ci*4*co/t/1e12:.1f} TFLOP/s  {gb/t/1e12:.2f} TB/s algorithmic  {digest}")
    elif kind == "attn":
        B, N, c = a[:3]
        iters = a[3] if len(a) > 3 else 10
        dq = max(8, c // 8)
        torch.manual_seed(1)
        qkv = torch.randn(B, N, 2 * dq + c, device=dev) * 0.5
        torch.manual_seed(0)
        x, dy = torch.randn(B, N, c, device=dev), torch.randn(B, N, c, device=dev)
        g = torch.full((1,), 0.5, device=dev)
        out, raw, lse = torch.empty_like(x), torch.empty_like(x), torch.empty(B, N, device=dev)
        t = timeit(lambda: call("ddpm_attention_fwd", P(qkv), P(x), c, P(g), P(out), c, P(raw), P(lse), None, B, N, c, dq, S()), iters)
        print(f"attn fwd B={B} N={N} C={c}: {t*1e6:.1f} us  {2.0*B*N*N*(dq+c)/t/1e12:.1f} TFLOP/s (training call: raw output and log-sum-exp)")
        t = timeit(lambda: call("ddpm_attention_fwd", P(qkv), P(x), c, P(g), P(out), c, None, None, None, B, N, c, dq, S()), iters)
        print(f"attn fwd B={B} N={N} C={c}: {t*1e6:.1f} us  {2.0*B*N*N*(dq+c)/t/1e12:.1f} TFLOP/s (inference call)")
        ws, es = torch.empty(4096, dtype=torch.float64, device=dev), torch.empty(B, N, device=dev)
        dqkv, dg = torch.empty_like(qkv), torch.empty(1, device=dev)
        t = timeit(lambda: call("ddpm_attention_bwd", P(qkv), P(dy), c, P(raw), P(lse), P(g), P(ws), P(es), P(dqkv), P(dg), B, N, c, dq,
                                S()), iters)
        import hashlib
        digest = hashlib.sha1(dqkv.cpu().numpy().tobytes() + out.cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"attn bwd B={B} N={N} C={c}: {t*1e6:.1f} us  {2.0*B*N*N*(3*dq+3*c)/t/1e12:.1f} TFLOP/s (algorithmic, no recompute)  outputs {digest}")
    elif kind == "attnblock":
        import hashlib
        B, N, c = a[:3]
        iters = a[3] if len(a) > 3 else 10
        dq = c // 8
        dqp = (dq + 7) // 8 * 8
        lq = 2 * dqp + c
        torch.manual_seed(2)
        x = torch.randn(B, N, c, device=dev)
        ws_ = [torch.randn(r, c, device=dev) * c ** -0.5 for r in (dq, dq, c)]
        bs_ = [torch.randn(r, device=dev) * 0.1 for r in (dq, dq, c)]
        wf, wt = torch.empty(packed_floats(1, lq, c), device=dev), torch.empty(packed_floats(1, c, lq), device=dev)
        bias, g = torch.empty(lq, device=dev), torch.full((1,), 0.5, device=dev)
        call("ddpm_pack_qkv", *(P(t) for t in ws_), *(P(t) for t in bs_), c, dq, dqp, P(wf), P(wt), P(bias), S())
        out, ws = torch.empty(B, N, c, device=dev), torch.empty(B, N, lq, device=dev)
        t = timeit(lambda: call("ddpm_self_attention_infer", P(x), c, P(wf), P(bias), P(g), None, P(out), c, P(ws), B, N, c, dqp, S()), iters)
        digest = hashlib.sha1(out[:8].cpu().numpy().tobytes()).hexdigest()[:12]
        print(f"attention block B={B} N={N} C={c}: {t*1e6:.1f} us  {(2.0*B*N*lq*c + 2.0*B*N*N*(dqp+c))/t/1e12:.1f} TFLOP/s  {digest}")
    elif kind == "step":
        n = a[0]
        iters = a[1] if len(a) > 1 else 50
        x, e, z, o = (torch.randn(n, device=dev) for _ in range(4))
        t = timeit(lambda: call("ddpm_reverse_step", P(x), P(e), None, 0.0, P(z), 1.01, 0.02, 0.1, P(o), n, S()), iters)
        print(f"reverse_step n={n}: {t*1e6:.1f} us  {16.0*n/t/1e9:.1f} GB/s")
    elif kind == "bn":
        import hashlib
        B, H, W, c = a[:4]
        iters = a[4] if len(a) > 4 else 30
        rows = B * H * W
        torch.manual_seed(0)
        x, dy, res = (torch.randn(rows, c, device=dev) for _ in range(3))
        w, b = torch.rand(c, device=dev) + 0.5, torch.randn(c, device=dev) * 0.1
        temb = torch.randn(B, c, device=dev)
        mean, invstd, rm, rv = (torch.zeros(c, device=dev) for _ in range(4))
        ws = torch.empty(8192 * c, dtype=torch.float64, device=dev)
        out, dx, dres = torch.empty_like(x), torch.empty_like(x), torch.empty_like(x)
        sums, dw, db, dcb = torch.empty(2 * c, device=dev), torch.empty(c, device=dev), torch.empty(c, device=dev), torch.empty(c, device=dev)
        mb = rows * c * 4 / 1e6
        digest = lambda *ts: hashlib.sha1(b"".join(t.cpu().numpy().tobytes() for t in ts)).hexdigest()[:12]
        t = timeit(lambda: call("ddpm_bn_stats", P(x), c, rows, c, 1e-5, 0.1, P(ws), P(mean), P(invstd), P(rm), P(rv), S()), iters)
        print(f"bn stats   {rows}x{c}: {t*1e6:6.1f} us  {mb/t/1e6:.2f} TB/s ({mb:.0f} MB per tensor)  {digest(mean, invstd)}")
        for act, r in ((2, None), (2, res)):
            k = 2 + (r is not None)
            t = timeit(lambda: call("ddpm_bn_apply_fwd", P(x), c, P(mean), P(invstd), P(w), P(b), P(r), c, P(temb), act, P(out), c, B, H * W, c,
                                    S()), iters)
            print(f"bn apply   {rows}x{c} res={r is not None}: {t*1e6:6.1f} us  {k*mb/t/1e6:.2f} TB/s  {digest(out)}")
            t = timeit(lambda: call("ddpm_bn_bwd", P(x), c, P(dy), c, P(r), c, P(mean), P(invstd), P(w), P(b), act, rows, c, P(ws), P(sums), P(dw),
                                    P(db), 0, P(dx), c, P(dres) if r is not None else None, c, P(dcb), S()), iters)
            k = 2 * (2 + (r is not None)) + 1 + (r is not None)
            print(f"bn bwd     {rows}x{c} res={r is not None}: {t*1e6:6.1f} us  {k*mb/t/1e6:.2f} TB/s (sums pass + apply pass)  "
                  f"{digest(dx, dw, db, dcb, *([dres] if r is not None else []))}")
    else:
        sys.exit(__doc__)


if __name__ == "__main__":
    main()
