#!/usr/bin/env python3
"""One complete DiffusionModel.sample() call (T = 1000 ancestral steps, the ten stage copies to the host
included) on the sampling bench's model: images/s = batch / wall time of the call."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
import diffusion_model_amd as pkg  # noqa: E402

wl = bench.WORKLOADS["sample"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else wl["batch"]
dev = torch.device("cuda", 0)
model = bench.build_model(wl["variant"], wl["cin"], wl["labels"]).to(dev).eval()
betas = pkg.BetaScheduler(1000, device=dev).get_schedule("linear")
dm = pkg.make_diffusion_class(wl["variant"])(model, betas, 1000)
labels = torch.tensor([3], device=dev)
with torch.no_grad():
    dm.sample((8, wl["cin"], wl["hw"], wl["hw"]), device=dev, labels=labels)      # warm-up (first-use initialisation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    stages = dm.sample((B, wl["cin"], wl["hw"], wl["hw"]), device=dev, labels=labels)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
x = stages[-1][0]
print(f"sample(): batch {B}, {len(stages)} stages, {dt:.2f} s -> {B / dt:.2f} images/s; final stage finite: {bool(torch.isfinite(x).all())}, "
      f"mean {x.mean().item():.3f} std {x.std().item():.3f}")
