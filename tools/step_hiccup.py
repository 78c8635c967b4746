#!/usr/bin/env python3
"""Phase times and allocator activity of the first training steps (looks for one-off stalls)."""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import bench  # noqa: E402
import diffusion_model_amd as pkg  # noqa: E402

wl = bench.WORKLOADS["train"]
dev = torch.device("cuda", 0)
model = bench.build_model(wl["variant"], wl["cin"], wl["labels"]).to(dev)
betas = pkg.BetaScheduler(1000, device=dev).get_schedule("linear")
dm = pkg.make_diffusion_class(wl["variant"])(model, betas, 1000)
model.train()
opt = torch.optim.Adam(model.parameters(), lr=2e-4)
crit = torch.nn.MSELoss()
B = wl["batch"]
data = torch.rand(B, 1, 28, 28, device=dev) * 2 - 1
keys = ("num_device_alloc", "num_device_free", "num_alloc_retries")
for i in range(14):
    s0 = {k: torch.cuda.memory_stats().get(k, 0) for k in keys}
    t = [time.perf_counter()]
    noise = torch.randn_like(data)
    ts = torch.randint(0, 1000, (B,), device=dev)
    pred = model(dm.add_noise(data, noise, ts), ts)
    loss = crit(pred, noise)
    t.append(time.perf_counter())
    opt.zero_grad()
    loss.backward()
    t.append(time.perf_counter())
    opt.step()
    t.append(time.perf_counter())
    loss.item()
    t.append(time.perf_counter())
    s1 = {k: torch.cuda.memory_stats().get(k, 0) for k in keys}
    d = [1e3 * (b - a) for a, b in zip(t, t[1:])]
    print(f"step {i}: fwd enqueue {d[0]:.1f} bwd enqueue {d[1]:.1f} adam enqueue {d[2]:.1f} wait {d[3]:.1f} | "
          + " ".join(f"{k}+{s1[k] - s0[k]}" for k in keys) + f" reserved {torch.cuda.memory_reserved() >> 20} MB")
