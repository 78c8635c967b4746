#!/usr/bin/env python3
"""Wall time of every training step from a cold start (each step ends in loss.item(), as in the bench)."""
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
times = []
for i in range(int(sys.argv[1]) if len(sys.argv) > 1 else 60):
    t0 = time.perf_counter()
    noise = torch.randn_like(data)
    t = torch.randint(0, 1000, (B,), device=dev)
    pred = model(dm.add_noise(data, noise, t), t)
    loss = crit(pred, noise)
    opt.zero_grad()
    loss.backward()
    opt.step()
    loss.item()
    times.append(1e3 * (time.perf_counter() - t0))
print(" ".join(f"{x:.1f}" for x in times))

# where a slow step comes from: garbage collections and device allocations per step
import gc
gc_log = []
def _cb(phase, info):
    if phase == "start":
        _cb.t = time.perf_counter()
    else:
        gc_log.append((info["generation"], 1e3 * (time.perf_counter() - _cb.t), info["collected"]))
gc.callbacks.append(_cb)
rows = []
for i in range(30):
    n0 = torch.cuda.memory_stats().get("num_device_alloc", 0)
    g0 = len(gc_log)
    t0 = time.perf_counter()
    noise = torch.randn_like(data)
    t = torch.randint(0, 1000, (B,), device=dev)
    pred = model(dm.add_noise(data, noise, t), t)
    loss = crit(pred, noise)
    opt.zero_grad()
    loss.backward()
    opt.step()
    loss.item()
    dt = 1e3 * (time.perf_counter() - t0)
    rows.append((dt, torch.cuda.memory_stats().get("num_device_alloc", 0) - n0, gc_log[g0:]))
for i, (dt, na, g) in enumerate(rows):
    if dt > 30 or g:
        print(f"step {i}: {dt:.1f} ms, {na} device allocations, gc {[(a, round(b, 1), c) for a, b, c in g]}")
print("reserved MB", torch.cuda.memory_reserved() >> 20, "allocated MB", torch.cuda.memory_allocated() >> 20)
