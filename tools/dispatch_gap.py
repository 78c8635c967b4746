#!/usr/bin/env python3
"""Cost of one dependent dispatch in a stream: N launches of a tiny kernel back to back (queue kept full)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from gpu_util import call, P, stream as S

x = torch.randn(64, 64, device="cuda")
ws = torch.empty(8192 * 64, dtype=torch.float64, device="cuda")
out = torch.empty(64, device="cuda")
n = 2000
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call("ddpm_colsum", P(x), 64, 1, 64, 64, P(ws), 0, P(out), S())     # two tiny kernels
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"colsum x{n}: host enqueue {1e6*(t1-t0)/n:.1f} us/call, total {1e6*(t2-t0)/(2*n):.2f} us per dispatch")
y = torch.empty_like(x)
for rep in range(2):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        y.copy_(x)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"torch copy_ x{n}: {1e6*(t2-t0)/n:.2f} us per dispatch")

# GPU-bound variant: a 20-30 us kernel launched back to back; per-launch time minus the kernel's own
# duration (take it from a rocprofv3 --kernel-trace run of this script) is the gap between dependent dispatches.
B, HW, Cc = 256, 784, 64
xx = torch.randn(B * HW, Cc, device="cuda")
oo = torch.empty_like(xx)
mean, inv, w, b = (torch.randn(Cc, device="cuda") for _ in range(4))
n = 3000
for rep in range(3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n):
        call("ddpm_bn_apply_fwd", P(xx), Cc, P(mean), P(inv), P(w), P(b), None, 0, None, 2, P(oo), Cc, B, HW, Cc, S())
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"bn_apply_fwd x{n}: host enqueue {1e6*(t1-t0)/n:.1f} us/call, {1e6*(t2-t0)/n:.2f} us per launch end to end")
