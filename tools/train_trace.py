#!/usr/bin/env python3
"""Loss trace of N training steps of the training bench's model on fixed synthetic data (seeded noise and timesteps per
step), for the arithmetic chosen by DDPM_CONV_MATH: the traces of the three arithmetics should lie on top of each other
for as long as fp32 rounding noise has not been amplified by the optimisation itself.
    python tools/train_trace.py [steps]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
import diffusion_model_amd as pkg  # noqa: E402
from diffusion_model_amd.optim import Adam  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 60
wl = bench.WORKLOADS["train"]
dev = torch.device("cuda", 0)
model = bench.build_model(wl["variant"], wl["cin"], wl["labels"]).to(dev).train()
betas = pkg.BetaScheduler(1000, device=dev).get_schedule("linear")
dm = pkg.make_diffusion_class(wl["variant"])(model, betas, 1000)
opt = Adam(model.parameters(), lr=2e-4)
gen = torch.Generator(device="cpu").manual_seed(7)
B, hw = 256, wl["hw"]
data = (torch.rand((B, 1, hw, hw), generator=gen) * 2 - 1).to(dev)
crit = torch.nn.MSELoss()
out = []
for s in range(steps):
    noise = torch.randn((B, 1, hw, hw), generator=gen).to(dev)
    t = torch.randint(0, 1000, (B,), generator=gen).to(dev)
    loss = crit(model(dm.add_noise(data, noise, t), t), noise)
    opt.zero_grad()
    loss.backward()
    opt.step()
    if s % 5 == 0 or s == steps - 1:
        out.append(f"{s}: {loss.item():.5f}")
print(f"[{os.environ.get('DDPM_CONV_MATH', 'f16x2 (default)')}] " + "  ".join(out))
