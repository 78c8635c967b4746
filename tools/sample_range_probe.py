#!/usr/bin/env python3
"""Magnitudes along one complete sample() call of the sampling bench's (randomly initialised) model: the largest |x| of
every stage the call returns, for the arithmetic chosen by DDPM_CONV_MATH. The fp16-pair arithmetic needs activations
below 65504; a trained model keeps x near unit scale, an untrained one need not."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
import diffusion_model_amd as pkg  # noqa: E402

wl = bench.WORKLOADS["sample"]
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
T = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
dev = torch.device("cuda", 0)
model = bench.build_model(wl["variant"], wl["cin"], wl["labels"]).to(dev).eval()
betas = pkg.BetaScheduler(T, device=dev).get_schedule("linear")
dm = pkg.make_diffusion_class(wl["variant"])(model, betas, T)
torch.manual_seed(0)
with torch.no_grad():
    stages = dm.sample((B, wl["cin"], wl["hw"], wl["hw"]), device=dev, labels=torch.tensor([3], device=dev))
for x, t in stages:
    print(f"t = {t:4d}: max |x| = {x.abs().max().item():.4g}, finite: {bool(torch.isfinite(x).all())}")
    fm = None
