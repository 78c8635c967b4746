"""Rehearsal probe: 2 ranks on one GPU, gloo backend, CUDA tensors (prints every phase)."""
import os, sys, time, torch, torch.distributed as dist
r = int(os.environ["RANK"])
def say(m): print(f"[{r} {time.strftime('%H:%M:%S')}] {m}", flush=True)
torch.cuda.set_device(0)
dist.init_process_group("gloo", rank=r, world_size=int(os.environ["WORLD_SIZE"]))
say("init")
parts = [torch.ones(n, device="cuda") for n in (1000, 5_000_000, 7_000_000, 3)]
for i in range(4):
    flat = torch.cat([p.reshape(-1) for p in parts])
    dist.all_reduce(flat); flat.div_(2); torch.cuda.synchronize(); say(f"fresh allreduce {i} ok {flat[0].item()}")
    if i == 1:
        dist.barrier(); say("barrier ok")
    if i >= 2:
        evs = []
        for k in range(600):
            e = torch.cuda.Event(enable_timing=True); e.record(); evs.append(e)
        say("events recorded")
dist.destroy_process_group(); say("done")
