#!/usr/bin/env python3
"""Does one UNet inference call capture into a HIP graph, does the replay equal the eager result, and
what does a sampling step cost each way at small batch?   python tools/graph_probe.py [variant] [hw] B..."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import diffusion_model_amd as pkg  # noqa: E402


def main():
    variant, hw = int(sys.argv[1]), int(sys.argv[2])
    U = pkg.make_unet_class(variant)
    torch.manual_seed(0)
    m = (U(1, 1, 64, 128) if variant < 3 else U(1, 1, 64, 128, num_classes=10)).cuda().eval()
    for B in [int(v) for v in sys.argv[3:]]:
        x = torch.randn(B, 1, hw, hw, device="cuda")
        t = torch.full((B,), 500, dtype=torch.long, device="cuda")
        lab = torch.zeros(B, dtype=torch.long, device="cuda") if variant == 3 else None
        args = (x, t, lab) if variant == 3 else (x, t)
        with torch.no_grad():
            for _ in range(3):
                ref = m(*args).clone()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(50):
                m(*args)
            torch.cuda.synchronize()
            eager = (time.perf_counter() - t0) / 50
            g = torch.cuda.CUDAGraph()
            s = torch.cuda.Stream()
            s.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(s):
                m(*args)
            torch.cuda.current_stream().wait_stream(s)
            with torch.cuda.graph(g):
                out = m(*args)
            g.replay()
            torch.cuda.synchronize()
            same = torch.equal(out, ref)
            x.copy_(torch.randn_like(x))
            ref2 = m(*args).clone()
            g.replay()
            torch.cuda.synchronize()
            same2 = torch.equal(out, ref2)
            t0 = time.perf_counter()
            for _ in range(50):
                g.replay()
            torch.cuda.synchronize()
            graph = (time.perf_counter() - t0) / 50
        print(f"v{variant} {hw}x{hw} B={B}: eager {eager*1e3:.3f} ms  graph {graph*1e3:.3f} ms  replay==eager {same} {same2}", flush=True)


if __name__ == "__main__":
    main()
