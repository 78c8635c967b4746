#!/usr/bin/env python3
"""Host-side cost of one training step: wall time of the enqueue phase and a cProfile of it."""
import cProfile
import os
import pstats
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


def step(sync=True):
    noise = torch.randn_like(data)
    t = torch.randint(0, 1000, (B,), device=dev)
    noisy = dm.add_noise(data, noise, t)
    pred = model(noisy, t)
    loss = crit(pred, noise)
    opt.zero_grad()
    loss.backward()
    opt.step()
    return loss.item() if sync else loss


for _ in range(3):
    step()
torch.cuda.synchronize()
t0 = time.perf_counter()
step(sync=False)
t1 = time.perf_counter()
torch.cuda.synchronize()
t2 = time.perf_counter()
print(f"enqueue {1e3 * (t1 - t0):.2f} ms, until GPU idle {1e3 * (t2 - t0):.2f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)


def phased():
    """One step with a CUDA event and a host timestamp at every phase boundary."""
    names, evs, host = [], [], []

    def mark(nm):
        e = torch.cuda.Event(enable_timing=True)
        e.record()
        names.append(nm), evs.append(e), host.append(time.perf_counter())

    mark("start")
    noise = torch.randn_like(data)
    t = torch.randint(0, 1000, (B,), device=dev)
    noisy = dm.add_noise(data, noise, t)
    mark("noise")
    pred = model(noisy, t)
    mark("forward")
    loss = crit(pred, noise)
    opt.zero_grad()
    mark("loss")
    loss.backward()
    mark("backward")
    opt.step()
    mark("adam")
    v = loss.item()
    mark("item")
    torch.cuda.synchronize()
    return names, evs, host, v


for _ in range(2):
    phased()
print("phase      host_ms(cum)  gpu_ms(phase)")
for rep in range(3):
    names, evs, host, _ = phased()
    for i in range(1, len(names)):
        print(f"{names[i]:10s} {1e3 * (host[i] - host[0]):10.2f} {evs[i - 1].elapsed_time(evs[i]):12.2f}")
    print(f"total gpu {evs[0].elapsed_time(evs[-1]):.2f} ms")
