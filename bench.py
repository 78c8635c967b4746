#!/usr/bin/env python3
"""Benchmark of the DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload train|sample] [--batch B]

One "step" is one pass of the hot path over one batch of synthetic input:
  train  (default; BASELINE.json configs[1]): the body of the reference's training loop
         (first-iteration/main.py:65-87) -- noise + timestep draw, forward noising, UNet train-mode
         forward, eps-MSE, backward, Adam step, loss.item() -- second-iteration UNet (1,1,64,128),
         1x28x28, batch 256 per GPU. value = train samples/s.
  sample (BASELINE.json configs[2]): one ancestral denoising step at batch B per GPU (UNet eval
         forward, noise draw, fused reverse-step kernel), third-iteration UNet, 1x32x32. The metric,
         sampled images/s over the full T=1000-step loop, is B_total / (1000 * seconds per step).
For N > 1 the batch axis is sharded (weak scaling: per-GPU batch fixed): no data-path collective for
sampling; an RCCL all-reduce (average) of the flattened gradients per training step.
Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3     # gfx950 dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2516.6    # gfx950 dense bf16 matrix peak (v_mfma_f32_32x32x16_bf16)
SPLIT_PRODUCTS = 6                # bf16 MFMA products per fp32 product in the split-bf16 kernels
HBM_PEAK_GBPS = 8000.0            # HBM3E peak of one MI355X
SUSTAINED_BF16_TFLOPS = 1900.0    # measured: pure-MFMA loop, two waves per SIMD (tools/probe/mfma_power.hip)


def kernel_peak(label):
    """Matrix-core ceiling of a kernel family in algorithmic (fp32) TFLOP/s."""
    if "split-bf16" in label:
        return BF16_MFMA_PEAK_TFLOPS / SPLIT_PRODUCTS
    return FP32_MFMA_PEAK_TFLOPS

T_FULL = 1000

WORKLOADS = {
    "train": dict(name="configs[1]: second-iteration UNet (1,1,64,128), MNIST-shaped 1x28x28, batch 256 training step "
                       "(noise, add_noise, fwd, eps-MSE, bwd, Adam)", variant=2, cin=1, hw=28, batch=256, labels=False,
                  metric="train samples/sec", unit="samples/s"),
    "sample": dict(name="configs[2]: third-iteration UNet (1,1,64,128,num_classes=10), 1x32x32, full T=1000 ancestral sampling, "
                        "batch 4096 sharded over 8 GPUs (512 per GPU)", variant=3, cin=1, hw=32, batch=512, labels=True,
                   metric="sampled images/sec (full T-step loop)", unit="images/s"),
    # the remaining configs: parity-test shapes, measurable here on request (never the default line)
    "train_color": dict(name="configs[3]: third-iteration UNet (3,3,64,128), colour 3x32x32, batch 256 per GPU training step, "
                             "data-parallel gradient all-reduce", variant=3, cin=3, hw=32, batch=256, labels=False,
                        metric="train samples/sec", unit="samples/s"),
    "sample_64": dict(name="configs[4]: third-iteration UNet (3,3,64,128), 3x64x64 (attention at N=4096, 1024, 256), full T=1000 "
                           "ancestral sampling, batch 8192 sharded over 8 GPUs (1024 per GPU)", variant=3, cin=3, hw=64, batch=1024,
                      labels=False, metric="sampled images/sec (full T-step loop)", unit="images/s"),
}


def build_model(variant, cin, labels):
    import diffusion_model_amd as pkg
    from seeding import seeded_state_dict
    U = pkg.make_unet_class(variant)
    torch.manual_seed(0)
    m = U(cin, cin, 64, 128, num_classes=10) if (variant == 3 and labels) else U(cin, cin, 64, 128)
    m.load_state_dict(seeded_state_dict(m.state_dict(), 0))      # non-trivial BN statistics and attention gates
    return m


def reverse_step_bandwidth(dev, loop_elements, big=1 << 26, iters=20):
    """The fused reverse-step kernel (x_{t-1} from x_t, eps-hat, z: 16 algorithmic bytes per element) against the HBM
    peak, timed with HIP events outside the timed steps: at 2^26 elements, where every operand comes from HBM, and at
    the size one denoising step of this workload launches it with (small enough to stay in the caches)."""
    import ctypes as C
    from diffusion_model_amd._lib import call
    res = {"kernel": "reverse_step", "bytes_per_element": 16, "peak": HBM_PEAK_GBPS, "unit": "GB/s"}
    for tag, n in (("hbm_resident", big), ("loop_size", loop_elements)):
        x, eps, z, out = (torch.randn(n, device=dev) for _ in range(4))
        st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
        fn = lambda: call("ddpm_reverse_step", C.c_void_p(x.data_ptr()), C.c_void_p(eps.data_ptr()), None, 0.0,  # noqa: E731
                          C.c_void_p(z.data_ptr()), 1.01, 0.02, 0.1, C.c_void_p(out.data_ptr()), n, st)
        for _ in range(3):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / iters
        gbps = 16.0 * n / (ms * 1e-3) / 1e9
        res[tag] = {"elements": n, "avg_ms": round(ms, 4), "achieved": round(gbps, 1), "frac": round(gbps / HBM_PEAK_GBPS, 4)}
        del x, eps, z, out
    return res


def cpu_baseline(workload, wl, seconds=12.0):
    """The CPU oracle (a torch-CPU restatement of the reference) timed on the host cores, bounded sample."""
    from oracle import ddpm_oracle as O
    m = build_model(wl["variant"], wl["cin"], wl["labels"])
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    betas = O.schedule_betas("linear", T_FULL)
    alphas, abar = O.schedule_tables(betas)
    B, cin, hw = 8, wl["cin"], wl["hw"]
    n, t0 = 0, time.time()
    if workload.startswith("sample"):
        x = torch.randn(B, cin, hw, hw)
        t = torch.full((B,), 500, dtype=torch.long)
        y = torch.tensor([3]) if wl["labels"] else None
        c = O.step_coefficients(betas, alphas, abar, 500)
        with torch.no_grad():
            while n < 2 or time.time() - t0 < seconds:
                eps = O.unet_forward(sd, wl["variant"], x, t, y, train=False)[0]
                x = torch.from_numpy(O.reverse_step(x.numpy(), eps.numpy(), torch.randn_like(x).numpy(), *c))
                n += 1
        dt = (time.time() - t0) / n
        return dict(value=B / (T_FULL * dt), unit="images/s", cores=torch.get_num_threads(), kind="port",
                    sample=f"{n} denoising steps of the CPU oracle at batch {B}, extrapolated to the {T_FULL}-step loop")
    params = {k: v.clone() for k, v in sd.items() if O.is_param(k)}
    state, cur = {}, dict(sd)
    while n < 2 or time.time() - t0 < seconds:
        x0, noise = torch.rand(B, cin, hw, hw) * 2 - 1, torch.randn(B, cin, hw, hw)
        t = torch.randint(0, T_FULL, (B,))
        noisy = torch.from_numpy(O.add_noise(x0.numpy(), noise.numpy(), t.numpy(), abar))
        cur.update(params)
        _, grads, _, bufs = O.loss_and_grads(cur, wl["variant"], noisy, t, noise)
        cur.update(bufs)
        O.adam_step(params, grads, state)
        n += 1
    dt = (time.time() - t0) / n
    return dict(value=B / dt, unit="samples/s", cores=torch.get_num_threads(), kind="port",
                sample=f"{n} training steps of the CPU oracle (torch-CPU autograd + Adam) at batch {B}")


def log(msg, trace=False):
    """Progress on stderr (stdout carries exactly one JSON line). trace messages (every rank) only
    when DDPM_BENCH_TRACE=<seconds> is set, which also arms a stack dump after that many seconds."""
    if trace:
        if not os.environ.get("DDPM_BENCH_TRACE"):
            return
    elif int(os.environ.get("RANK", "0")) != 0:
        return
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="train", choices=list(WORKLOADS))
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--optimizer", default="hip", choices=["hip", "torch"], help="training: diffusion_model_amd.optim.Adam or torch.optim.Adam")
    ap.add_argument("--profile-stride", type=int, default=7,
                    help="bracket every N-th MFMA launch of the timed steps with HIP events (1 = all; events cost GPU time)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--settle", type=float, default=1.5, help="seconds of untimed steps before the warm-up steps")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the multi-GPU code path (process group, collectives) "
                                                              "even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world == 1 and args.gpus > 1:
        sys.exit("launch with `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...`")
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist = None
    multi = world > 1 or args.force_dist
    log(f"world={world} device={dev} workload={args.workload}")
    if os.environ.get("DDPM_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DDPM_BENCH_TRACE"]), exit=True)
    import diffusion_model_amd as pkg
    from diffusion_model_amd import parallel
    from diffusion_model_amd._lib import call
    from diffusion_model_amd.schedule import step_coefficients_host
    wl = WORKLOADS[args.workload]
    B = args.batch or wl["batch"]
    shape = (B, wl["cin"], wl["hw"], wl["hw"])
    model = build_model(wl["variant"], wl["cin"], wl["labels"]).to(dev)
    betas = pkg.BetaScheduler(T_FULL, device=dev).get_schedule("linear")
    dm = pkg.make_diffusion_class(wl["variant"])(model, betas, T_FULL)
    gen = torch.Generator(device="cpu").manual_seed(1234 + rank)

    if args.workload.startswith("sample"):
        model.eval()
        labels = torch.tensor([3], device=dev) if wl["labels"] else None
        state = {"x": torch.randn(shape, generator=gen).to(dev), "t": T_FULL - 1}

        def one_step():
            """Body of the sampling loop for one timestep (third-iteration/diffusion.py:24-44)."""
            x, t = state["x"], state["t"]
            with torch.no_grad():
                tt = torch.full((B,), t, dtype=torch.long, device=dev)
                eps = model(x, tt, labels) if wl["variant"] == 3 else model(x, tt)
                c1, c2, c3 = step_coefficients_host(dm._betas_h[t], dm._alphas_h[t], dm._abar_h[t])
                z = torch.randn_like(x)
                out = torch.empty_like(x)
                call("ddpm_reverse_step", C.c_void_p(x.data_ptr()), C.c_void_p(eps.data_ptr()), None, 0.0, C.c_void_p(z.data_ptr()),
                     c1, c2, c3, C.c_void_p(out.data_ptr()), x.numel(), C.c_void_p(torch.cuda.current_stream().cuda_stream))
            state["x"], state["t"] = out, t - 1
    else:
        model.train()
        from diffusion_model_amd.optim import Adam
        opt = (torch.optim.Adam if args.optimizer == "torch" else Adam)(model.parameters(), lr=2e-4)
        crit = torch.nn.MSELoss()
        data = (torch.rand(shape, generator=gen) * 2 - 1).to(dev)            # synthetic MNIST-shaped batch in [-1, 1]
        params = [p for p in model.parameters()]
        state = {"loss": 0.0}

        def one_step():
            """Body of the reference's training loop (first-iteration/main.py:65-87)."""
            x = data
            noise = torch.randn_like(x)
            timesteps = torch.randint(0, T_FULL, (B,), device=dev)
            noisy = dm.add_noise(x, noise, timesteps)
            pred = model(noisy, timesteps)
            loss = crit(pred, noise)
            opt.zero_grad()
            loss.backward()
            if dist is not None and not os.environ.get("DDPM_BENCH_SKIP_ALLREDUCE"):   # data-parallel: average gradients over ranks
                parallel.average_gradients(params)
            elif os.environ.get("DDPM_BENCH_FLAT_GRADS"):                    # diagnostic: gradients as views of one buffer
                gs = [p for p in params if p.grad is not None]
                flat = torch.cat([p.grad.reshape(-1) for p in gs])
                off = 0
                for p in gs:
                    n = p.grad.numel()
                    p.grad = flat[off:off + n].view_as(p.grad)
                    off += n
            opt.step()
            if os.environ.get("DDPM_BENCH_DEFER_LOSS"):                      # diagnostic: no host sync inside the step
                state["loss_t"] = loss
            else:
                state["loss"] = loss.item()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            log(f"rank {rank}: entering barrier", True)
            if args.backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
            torch.cuda.synchronize()
            log(f"rank {rank}: left barrier", True)

    prof = None
    if multi:
        # Per-kernel HIP events are taken in a pre-pass BEFORE the process group exists (2 steps, no
        # collective), so that event recording never interleaves with RCCL traffic in the timed region.
        log("multi-GPU: profiling pre-pass (no collectives)")
        one_step()
        model._engine.prof = []
        one_step()
        one_step()
        torch.cuda.synchronize()
        prof, prof_dt = model._engine.prof, None
        model._engine.prof = None
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29533")
        if os.environ.get("DDPM_BENCH_NO_PG"):      # diagnostic: the multi-GPU control flow without a process group
            dist_mod = None
        elif args.backend == "nccl":
            dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist_mod.init_process_group(args.backend, rank=rank, world_size=world)
        dist = dist_mod
        if dist is not None:
            parallel.broadcast_parameters(model)

    # Settling steps before the W warm-up steps: the first second of a process holds one-off host stalls
    # (first-use initialisation inside the HIP runtime and the caching allocator: a single step of the first
    # dozen takes 50-80 ms instead of 29), which would otherwise land inside a short timed region.
    log("model built; settling")
    t_settle = time.perf_counter() + args.settle
    n_settle = 0
    if multi:
        # every rank must issue the same number of collectives: a fixed count instead of a time budget
        for _ in range(int(round(args.settle * 30))):
            one_step()
            n_settle += 1
    else:
        while time.perf_counter() < t_settle:
            one_step()
            n_settle += 1
    torch.cuda.synchronize()
    log(f"{n_settle} settling steps; warm-up")
    for _ in range(args.warmup):
        one_step()
    log("timed steps")
    if not multi:
        model._engine.prof = []
        model._engine.prof_stride = max(1, args.profile_stride)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        one_step()
        log(f"rank {rank}: step {i} issued", True)
    barrier()
    dt = time.perf_counter() - t0
    log(f"done: {dt / args.steps * 1e3:.2f} ms/step on this rank")
    if not multi:
        prof = model._engine.prof
        model._engine.prof = None
    if args.workload.startswith("sample"):
        assert torch.isfinite(state["x"]).all()
    else:
        assert state["loss"] == state["loss"], "loss is NaN"
    if dist is not None:
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        parallel.all_reduce_(tt, op=dist.ReduceOp.MAX)
        dt = tt.item()
    sec_per_step = dt / args.steps
    total_b = B * world
    value = total_b / (T_FULL * sec_per_step) if args.workload.startswith("sample") else total_b / sec_per_step

    # per-kernel figures from the HIP events recorded around every MFMA launch of the timed steps
    agg = {}
    for label, flops, e0, e1 in prof:
        a = agg.setdefault(label, [0.0, 0.0, 0])
        a[0] += flops
        a[1] += e0.elapsed_time(e1) * 1e-3
        a[2] += 1
    if os.environ.get("DDPM_BENCH_PER_SHAPE") and rank == 0:   # diagnostic: the same figures per distinct launch shape
        shapes = {}
        for label, flops, e0, e1 in prof:
            a = shapes.setdefault((label, flops), [0.0, 0])
            a[0] += e0.elapsed_time(e1) * 1e-3
            a[1] += 1
        for (label, flops), (t, n) in sorted(shapes.items(), key=lambda kv: -kv[1][0]):
            log(f"  {label:40s} {flops / 1e9:9.2f} GFLOP  x{n:3d}  {t / n * 1e6:8.1f} us  {flops * n / t / 1e12:6.1f} TFLOP/s")
    if not agg:
        sys.exit(f"no MFMA launch was sampled: lower --profile-stride ({sec_per_step * 1e3:.3f} ms/step)")
    dom = max(agg.items(), key=lambda kv: kv[1][1])
    achieved = dom[1][0] / dom[1][1] / 1e12
    # launches are sampled 1 in `stride` (all of them in the multi-GPU pre-pass): scale the summed time back up
    prof_steps = args.steps / max(1, args.profile_stride) if not multi else 2
    kernels = {k: dict(sampled_launches=v[2], avg_ms=round(v[1] / v[2] * 1e3, 4), tflops=round(v[0] / v[1] / 1e12, 2),
                       peak_tflops=round(kernel_peak(k), 1), frac=round(v[0] / v[1] / 1e12 / kernel_peak(k), 4),
                       share_of_step=round(v[1] / prof_steps / sec_per_step, 3)) for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1])}
    cfg = {"workload": wl["name"], "per_gpu_batch": B, "global_batch": total_b}
    if args.workload.startswith("sample"):
        cfg.update(T=T_FULL, step="one denoising step (UNet eval forward + noise draw + fused reverse-step kernel); "
                                  "images/s = global_batch / (T * seconds_per_step)")
    else:
        cfg.update(optimizer=("torch.optim.Adam(lr=2e-4)" if args.optimizer == "torch" else "Adam(lr=2e-4), torch.optim.Adam's update as one HIP launch per 64 tensors"), batchnorm="per-rank batch statistics",
                   parallelism=("dp%d: %s all-reduce of flattened gradients" % (world, "RCCL" if args.backend == "nccl" else args.backend))
                   if multi else "single GPU",
                   last_loss=round(state["loss"], 5))
    # memory-side bytes per launch of the dominant kernel: not measurable live (hardware counters need their own
    # rocprofv3 --pmc passes), so taken from the committed summary of tools/pmc_traffic.sh for this workload
    traffic, traffic_src = None, None
    kernel_symbol = {"conv3x3_halo<split-bf16>": "conv3x3_halo_kernel", "attention_fwd<split-bf16>": "attention_fwd_x3_kernel"}.get(dom[0])
    tfile = os.path.join(ROOT, "profiles", "r01_pmc_traffic_%s.txt" % ("sample" if args.workload == "sample" else "train"))
    if kernel_symbol and args.workload in ("train", "sample") and os.path.exists(tfile):
        for line in open(tfile):
            f = line.split()
            if len(f) > 4 and kernel_symbol in line and f[0].isdigit():
                traffic = int(float(f[3]) * 1e6)
                traffic_src = ("bytes per launch, mean over this kernel's launches of one step: rocprofv3 --pmc FETCH_SIZE (x 2, the gfx950 "
                               "correction for wide loads) + WRITE_SIZE in separate passes, profiles/" + os.path.basename(tfile))
                break
    out = {
        "metric": wl["metric"], "value": round(value, 4), "unit": wl["unit"],
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
        "config": cfg,
        "roofline": {"bound": "mfma", "kernel": dom[0], "achieved": round(achieved, 2), "peak": round(kernel_peak(dom[0]), 1),
                     "unit": "TFLOP/s", "frac": round(achieved / kernel_peak(dom[0]), 4), "traffic": traffic, "traffic_source": traffic_src,
                     "sustained_peak": (round(SUSTAINED_BF16_TFLOPS / SPLIT_PRODUCTS, 1) if "split-bf16" in dom[0] else None),
                     "sustained_peak_note": "a loop of nothing but independent v_mfma_f32_32x32x16_bf16 (tools/probe/mfma_power.hip) "
                                            "sustains 1.9 PFLOP/s on this chip (1.8 GHz under that load, not the 2.4 GHz of the nominal peak)",
                     "arithmetic": ("fp32 operands split exactly into 3 bf16 terms, 6 cross products per fp32 product on "
                                    "v_mfma_f32_32x32x16_bf16 with fp32 accumulation: achieved/peak count algorithmic fp32 FLOP, "
                                    "peak = bf16 dense peak / 6" if "split-bf16" in dom[0] else "v_mfma_f32_32x32x2_f32"),
                     "avg_launch_ms": round(dom[1][1] / dom[1][2] * 1e3, 4), "sampled_launches": dom[1][2],
                     "measured": f"HIP events around every {max(1, args.profile_stride)}-th MFMA launch of the timed steps (rotating over the launch sites)" if not multi else
                                 "HIP events in a 2-step pre-pass before the process group is created"},
        "kernels": kernels,
    }
    if rank == 0:
        if not multi:
            out["hbm_kernel"] = reverse_step_bandwidth(dev, B * wl["cin"] * wl["hw"] * wl["hw"])
        if not multi and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(args.workload, wl)
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
