#!/usr/bin/env python3
"""Benchmark of the DDPM hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W [--workload both|train|sample|train_color|sample_64|sample_c1] [--batch B]

One "step" is one pass of the hot path over one batch of synthetic input:
  train  (BASELINE.json configs[1]): the body of the reference's training loop
         (first-iteration/main.py:65-87) -- noise + timestep draw, forward noising, UNet train-mode
         forward, eps-MSE, backward, Adam step, loss.item() -- second-iteration UNet (1,1,64,128),
         1x28x28, batch 256 per GPU (weak scaling: the reference's loop at a larger global batch).
         value = train samples/s.
  sample (BASELINE.json configs[2]): ancestral sampling of a GLOBAL batch of 4096 images, third-iteration
         UNet, 1x32x32, sharded over the N GPUs (strong scaling: 4096 / N images per GPU). Two measurements:
         K timed denoising steps (UNet eval forward, noise draw, fused reverse-step kernel), and ONE real
         DiffusionModel.sample() call over the full T = 1000 steps, wall clock, its ten stage copies to
         the host included: value = sampled images/s = 4096 / that wall time.
The default (--workload both) measures BOTH halves of the metric: the JSON line's top-level fields are
the training half, its "sampling" object the sampling half. Each half also reports the same K steps in
the full-range arithmetic (three bf16 planes per operand) beside the default fp16 pair ("full_range").
For N > 1 the driver launches one rank per GPU with torch.distributed.run; started as plain
`python bench.py --gpus N` this script starts those N ranks itself, before anything touches a GPU. No
data-path collective for sampling; RCCL all-reduces of gradient buckets (launched while the backward is
still running) and of the BatchNorm sums (global-batch statistics: the reference's semantics) per
training step. Prints ONE JSON line on rank 0.
"""
import argparse
import ctypes as C
import json
import os
import platform
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, "tests", "golden")):
    if p not in sys.path:
        sys.path.insert(0, p)

FP32_MFMA_PEAK_TFLOPS = 157.3     # gfx950 dense fp32 matrix peak (v_mfma_f32_32x32x2_f32)
BF16_MFMA_PEAK_TFLOPS = 2516.6    # gfx950 dense bf16 / fp16 matrix peak (v_mfma_f32_32x32x16_bf16, ..._f16)
HBM_PEAK_GBPS = 8000.0            # HBM3E peak of one MI355X
T_FULL = 1000

# The two split arithmetics: 16-bit MFMA products per fp32 product, the rate a loop of nothing but independent MFMAs of
# that type sustains on this chip (two waves per SIMD, changing operands), and how to say it in the JSON line.
ARITHMETIC = {
    "split-bf16": dict(products=6, sustained=1900.0, probe="tools/probe/mfma_power.hip", inst="v_mfma_f32_32x32x16_bf16",
                       text="fp32 operands split exactly into 3 bf16 terms, 6 cross products per fp32 product on v_mfma_f32_32x32x16_bf16 "
                            "with fp32 accumulation: achieved/peak count algorithmic fp32 FLOP, peak = bf16 dense peak / 6"),
    "fp16-pair": dict(products=3, sustained=1640.0, probe="tools/probe/f16x2_probe.hip", inst="v_mfma_f32_32x32x16_f16",
                      text="fp32 operands as fp16 pairs (hi + remainder scaled by 2^11: 22-23 significant bits), 3 cross products per fp32 "
                           "product on v_mfma_f32_32x32x16_f16 with fp32 accumulation; error against fp64 below that of the fp32 MFMA "
                           "(tests/test_gpu_accuracy.py); operands beyond +-65504 raise a flag and the call is repeated with three bf16 "
                           "planes (tests/test_gpu_range.py): achieved/peak count algorithmic fp32 FLOP, peak = fp16 dense peak / 3"),
}


def arithmetic_of(label):
    return next((a for a in ARITHMETIC if a in label), None)


def kernel_peak(label):
    """Matrix-core ceiling of a kernel family in algorithmic (fp32) TFLOP/s."""
    a = arithmetic_of(label)
    return BF16_MFMA_PEAK_TFLOPS / ARITHMETIC[a]["products"] if a else FP32_MFMA_PEAK_TFLOPS


# batch: per GPU (weak scaling) for the training workloads; global_batch: over all GPUs (strong scaling) for sampling
WORKLOADS = {
    "train": dict(name="configs[1]: second-iteration UNet (1,1,64,128), MNIST-shaped 1x28x28, batch 256 per GPU training step "
                       "(noise, add_noise, fwd, eps-MSE, bwd, Adam)", variant=2, cin=1, hw=28, batch=256, labels=False,
                  metric="train samples/sec", unit="samples/s"),
    "sample": dict(name="configs[2]: third-iteration UNet (1,1,64,128,num_classes=10), 1x32x32, full T=1000 ancestral sampling of a "
                        "global batch of 4096 images sharded over the GPUs", variant=3, cin=1, hw=32, global_batch=4096, labels=True,
                   metric="sampled images/sec (full T-step loop)", unit="images/s"),
    # the remaining configs: parity-test shapes, measurable here on request (never the default line)
    "sample_c1": dict(name="configs[0]: first-iteration UNet (1,1,64,128), 1x28x28, batch 64, full T=1000 ancestral sampling with the "
                           "linear schedule (the case the reference runs on the CPU)", variant=1, cin=1, hw=28, global_batch=64, labels=False,
                      metric="sampled images/sec (full T-step loop)", unit="images/s"),
    "train_color": dict(name="configs[3]: third-iteration UNet (3,3,64,128), colour 3x32x32, batch 256 per GPU training step, "
                             "data-parallel gradient all-reduce", variant=3, cin=3, hw=32, batch=256, labels=False,
                        metric="train samples/sec", unit="samples/s"),
    "sample_64": dict(name="configs[4]: third-iteration UNet (3,3,64,128), 3x64x64 (attention at N=4096, 1024, 256), full T=1000 "
                           "ancestral sampling of a global batch of 8192 images sharded over the GPUs (8 GPUs: 1024 each)", variant=3, cin=3,
                      hw=64, global_batch=8192, labels=False, metric="sampled images/sec (full T-step loop)", unit="images/s"),
}


def build_model(variant, cin, labels, bounded_samples=False):
    """Reference-default initialisation under a fixed seed with non-trivial BatchNorm statistics and attention gates.
    bounded_samples: the output layer is scaled by 1/64. An UNTRAINED network's eps-hat has nothing to do with the noise
    in x_t, and with the default output layer a 1000-step trajectory grows to 4e7 (a trained network keeps x_t at unit
    scale); scaled, the trajectory stays within a few thousand, so that the whole call runs in the arithmetic a trained
    network would run in. Same layers, same shapes, same kernels, same FLOP."""
    import torch
    import diffusion_model_amd as pkg
    from seeding import seeded_state_dict
    U = pkg.make_unet_class(variant)
    torch.manual_seed(0)
    m = U(cin, cin, 64, 128, num_classes=10) if (variant == 3 and labels) else U(cin, cin, 64, 128)
    sd = seeded_state_dict(m.state_dict(), 0)
    if bounded_samples:
        sd["final_conv.weight"] = sd["final_conv.weight"] / 64.0
        sd["final_conv.bias"] = sd["final_conv.bias"] / 64.0
    m.load_state_dict(sd)
    return m


def log(msg, trace=False):
    """Progress on stderr (stdout carries exactly one JSON line). trace messages (every rank) only
    when DDPM_BENCH_TRACE=<seconds> is set, which also arms a stack dump after that many seconds."""
    if trace:
        if not os.environ.get("DDPM_BENCH_TRACE"):
            return
    elif int(os.environ.get("RANK", "0")) != 0:
        return
    print(f"[bench {time.strftime('%H:%M:%S')}] {msg}", file=sys.stderr, flush=True)


def reverse_step_bandwidth(dev, loop_elements, big=1 << 26, iters=20):
    """The fused reverse-step kernel (x_{t-1} from x_t, eps-hat, z: 16 algorithmic bytes per element) against the HBM
    peak, timed with HIP events outside the timed steps: at 2^26 elements, where every operand comes from HBM, and at
    the size one denoising step of this workload launches it with."""
    import torch
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
    # the kernel the sampling loop actually runs: final 1x1 conv (64 channels -> 1) fused with the update. Per output element it
    # reads the pixel's 64 features (256 B) + x_t (+ z) and writes x_{t-1}: 268 B with an injected z, 264 B with in-kernel noise;
    # the step's own share is 12 B (8 B) instead of the standalone kernel's 16 B + the 8 B eps-hat round trip it removes
    B, hw, c = 4096, 1024, 64
    feat, w, bias = torch.randn(B * hw, c, device=dev), torch.randn(1, c, device=dev), torch.zeros(1, device=dev)
    x, z = torch.randn(B * hw, device=dev), torch.randn(B * hw, device=dev)
    st = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    for tag, mode, nbytes in (("fused_final_step_injected_z", 1, 4 * c + 12), ("fused_final_step_in_kernel_noise", 2, 4 * c + 8)):
        fn = lambda: call("ddpm_final_conv_reverse_step", C.c_void_p(feat.data_ptr()), c, C.c_void_p(w.data_ptr()),  # noqa: E731
                          C.c_void_p(bias.data_ptr()), C.c_void_p(x.data_ptr()), None, 0.0, mode, C.c_void_p(z.data_ptr()), C.c_uint64(1),
                          C.c_uint64(2), 1.01, 0.02, 0.1, C.c_void_p(x.data_ptr()), None, B, 32, 32, c, 1, st)
        for _ in range(2):
            fn()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        gbps = nbytes * B * hw / (ms * 1e-3) / 1e9
        res[tag] = {"elements": B * hw, "bytes_per_element": nbytes, "avg_ms": round(ms, 4), "achieved": round(gbps, 1),
                    "frac": round(gbps / HBM_PEAK_GBPS, 4)}
    return res


def host_description():
    model = platform.processor() or "unknown"
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    return model


def cpu_baseline_one(workload, wl, batch, threads, min_steps=2, seconds=8.0):
    """The CPU oracle (a torch-CPU restatement of the reference) timed on `threads` host cores on a bounded sample of
    the workload: at least `min_steps` steps and about `seconds` seconds, at batch `batch` (the workload's own batch
    when a step of it fits that budget, a smaller one otherwise: the rate per image hardly depends on it)."""
    import torch
    from oracle import ddpm_oracle as O
    prev = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        m = build_model(wl["variant"], wl["cin"], wl["labels"])
        sd = {k: v.clone() for k, v in m.state_dict().items()}
        betas = O.schedule_betas("linear", T_FULL)
        alphas, abar = O.schedule_tables(betas)
        B, cin, hw = batch, wl["cin"], wl["hw"]
        common = dict(cores=torch.get_num_threads(), kind="port", cpu=host_description(), torch=torch.__version__, batch=B)
        n, t0 = 0, time.time()
        if workload.startswith("sample"):
            x = torch.randn(B, cin, hw, hw)
            t = torch.full((B,), 500, dtype=torch.long)
            y = torch.tensor([3]) if wl["labels"] else None
            c = O.step_coefficients(betas, alphas, abar, 500)
            with torch.no_grad():
                while n < min_steps or time.time() - t0 < seconds:
                    eps = O.unet_forward(sd, wl["variant"], x, t, y, train=False)[0]
                    x = torch.from_numpy(O.reverse_step(x.numpy(), eps.numpy(), torch.randn_like(x).numpy(), *c))
                    n += 1
            dt = (time.time() - t0) / n
            return dict(value=round(B / (T_FULL * dt), 5), unit="images/s", **common,
                        sample=f"{n} denoising steps of the CPU oracle at batch {B} ({dt:.2f} s each), extrapolated to the {T_FULL}-step loop")
        params = {k: v.clone() for k, v in sd.items() if O.is_param(k)}
        state, cur = {}, dict(sd)
        while n < min_steps or time.time() - t0 < seconds:
            x0, noise = torch.rand(B, cin, hw, hw) * 2 - 1, torch.randn(B, cin, hw, hw)
            t = torch.randint(0, T_FULL, (B,))
            noisy = torch.from_numpy(O.add_noise(x0.numpy(), noise.numpy(), t.numpy(), abar))
            cur.update(params)
            _, grads, _, bufs = O.loss_and_grads(cur, wl["variant"], noisy, t, noise)
            cur.update(bufs)
            O.adam_step(params, grads, state)
            n += 1
        dt = (time.time() - t0) / n
        return dict(value=round(B / dt, 3), unit="samples/s", **common,
                    sample=f"{n} training steps of the CPU oracle (torch-CPU autograd + Adam) at batch {B} ({dt:.2f} s each)")
    finally:
        torch.set_num_threads(prev)


def cpu_baseline(workload, wl, batch):
    """The baseline on the threads torch uses by default on this host (the box's CPU share) and, for comparison with
    other hosts, on 8 threads (under "threads_8")."""
    import torch
    b = min(batch, 256)            # a 4096-image step of the oracle would take minutes; the per-image rate is the same
    full = cpu_baseline_one(workload, wl, b, torch.get_num_threads())
    if torch.get_num_threads() != 8:
        full["threads_8"] = {k: v for k, v in cpu_baseline_one(workload, wl, b, 8, seconds=6.0).items() if k in ("value", "unit", "cores", "sample")}
    return full


class Job:
    """One workload on this rank: model, the step function, and what the measurement collects."""

    def __init__(self, workload, args, ctx):
        import torch
        import diffusion_model_amd as pkg
        from diffusion_model_amd import parallel
        from diffusion_model_amd.schedule import step_coefficients_host
        self.workload, self.args, self.ctx = workload, args, ctx
        self.wl = wl = WORKLOADS[workload]
        dev, rank, world = ctx["dev"], ctx["rank"], ctx["world"]
        self.sampling = workload.startswith("sample")
        if self.sampling:
            self.global_batch = args.batch * world if args.batch else wl["global_batch"]
            lo, hi = parallel.shard_range(self.global_batch, world, rank)
            self.B = B = hi - lo                       # this rank's shard of the global batch
        else:
            self.B = B = args.batch or wl["batch"]
            self.global_batch = B * world
        self.shape = shape = (B, wl["cin"], wl["hw"], wl["hw"])
        self.model = model = build_model(wl["variant"], wl["cin"], wl["labels"], bounded_samples=self.sampling).to(dev)
        betas = pkg.BetaScheduler(T_FULL, device=dev).get_schedule("linear")
        self.dm = dm = pkg.make_diffusion_class(wl["variant"])(model, betas, T_FULL)
        gen = torch.Generator(device="cpu").manual_seed(1234 + rank)
        torch.manual_seed(4321 + rank)      # build_model seeds every rank alike (same weights); the noise of the steps must differ per rank
        self.state = state = {}
        self.reducer = None
        if self.sampling:
            model.eval()
            self.labels = labels = torch.tensor([3], device=dev) if wl["labels"] else None
            state.update(x=torch.randn(shape, generator=gen).to(dev), t=T_FULL - 1)

            def one_step():
                """Body of the sampling loop for one timestep (third-iteration/diffusion.py:24-44)."""
                x, t = state["x"], state["t"]
                with torch.no_grad():
                    tt = torch.full((B,), t, dtype=torch.long, device=dev)
                    c1, c2, c3 = step_coefficients_host(dm._betas_h[t], dm._alphas_h[t], dm._abar_h[t])
                    if args.noise == "philox":          # drawn inside the fused last kernel
                        step = dict(c1=c1, c2=c2, c3=c3, noise_mode=2, z=None, seed=1234 + rank, call=t, t_uniform=True, want_fms=False)
                    else:                               # what sample() does by default: torch's device generator
                        step = dict(c1=c1, c2=c2, c3=c3, noise_mode=1, z=torch.randn_like(x), t_uniform=True, want_fms=False)
                    out = model.denoise_step(x, tt, labels, step)     # UNet forward whose last layer is fused with the update
                if t > 1:
                    state["x"], state["t"] = out, t - 1
                else:       # the schedule is used up: the next step starts another call's chain from fresh noise, as sample()
                    state["x"], state["t"] = torch.randn_like(out), T_FULL - 1      # does (x carried on would leave every
                                                                                      # range the schedule keeps it in)
        else:
            model.train()
            from diffusion_model_amd.optim import Adam
            opt = (torch.optim.Adam if args.optimizer == "torch" else Adam)(model.parameters(), lr=2e-4)
            crit = torch.nn.MSELoss()
            data = (torch.rand(shape, generator=gen) * 2 - 1).to(dev)            # synthetic image batch in [-1, 1]
            params = [p for p in model.parameters()]
            state["loss"] = 0.0

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
                if ctx["dist"] is not None:                                      # data-parallel: average gradients over ranks
                    if self.reducer is not None:
                        self.reducer.finish()                                    # buckets were launched during the backward
                    else:
                        parallel.average_gradients(params)
                opt.step()
                state["loss"] = loss.item()
        self.one_step = one_step
        self.prof = None

    # ---- phases -------------------------------------------------------------------------------
    def prepass(self):
        """Per-kernel HIP events for the multi-GPU case, taken BEFORE the process group exists (2 steps, no
        collective), so that event recording never interleaves with RCCL traffic in the timed region."""
        import torch
        self.one_step()
        self.model._engine.prof = []
        self.one_step()
        self.one_step()
        torch.cuda.synchronize()
        self.prof = self.model._engine.prof
        self.model._engine.prof = None

    def attach_distributed(self):
        from diffusion_model_amd import parallel
        if self.ctx["dist"] is None:
            return
        parallel.broadcast_parameters(self.model)
        if not self.sampling and self.args.allreduce == "bucketed":
            self.reducer = parallel.BucketedGradientAverager(self.model, bucket_bytes=self.args.bucket_mb << 20)
            self.reducer.attach()
        if not self.sampling and self.args.batchnorm == "sync":
            self.model._engine.sync_bn = True

    def settle(self):
        # Settling steps before the W warm-up steps: the first second of a process holds one-off host stalls
        # (first-use initialisation inside the HIP runtime and the caching allocator: a single step of the first
        # dozen takes 50-80 ms instead of 29), which would otherwise land inside a short timed region.
        import torch
        args, multi = self.args, self.ctx["multi"]
        n = 0
        if multi:     # every rank must issue the same number of collectives: a fixed count instead of a time budget
            for _ in range(max(2, int(round(args.settle * (8 if self.sampling else 30))))):
                self.one_step()
                n += 1
        else:
            t_end = time.perf_counter() + args.settle
            while time.perf_counter() < t_end or n < 2:
                self.one_step()
                n += 1
        torch.cuda.synchronize()
        return n

    def timed(self, barrier, profile=True):
        import torch
        args, multi = self.args, self.ctx["multi"]
        for _ in range(args.warmup):
            self.one_step()
        if not multi and profile:
            self.model._engine.prof = []
            self.model._engine.prof_stride = max(1, args.profile_stride)
        barrier()
        t0 = time.perf_counter()
        for i in range(args.steps):
            self.one_step()
            log(f"rank {self.ctx['rank']}: step {i} issued", True)
        barrier()
        dt = time.perf_counter() - t0
        if not multi and profile:
            self.prof = self.model._engine.prof
            self.model._engine.prof = None
        if self.sampling:
            assert torch.isfinite(self.state["x"]).all()
        else:
            assert self.state["loss"] == self.state["loss"], "loss is NaN"
        return dt

    def full_range_steps(self, barrier):
        """The same K steps with three bf16 planes per operand (no range limit) instead of the fp16 pair."""
        from diffusion_model_amd._lib import conv_math
        eng = self.model._engine
        if eng.effective_math() != "f16x2":
            return None
        before = eng.math
        eng.math = "bf16x3"
        try:
            self.one_step()
            self.one_step()                      # packs the weights in the other format, sizes the allocator's pools
            dt = self.timed(barrier, profile=False)
        finally:
            eng.math = before
            self.one_step()                      # back on the default arithmetic (packed weights of both modes stay cached)
        return dt

    def full_call(self, barrier, T=T_FULL):
        """ONE real DiffusionModel.sample() call of this rank's shard over the full T-step schedule, wall clock between
        two barriers, the ten stage copies to the host included: the quantity the metric names."""
        import torch
        dev = self.ctx["dev"]
        dm = self.dm
        if T != T_FULL:
            import diffusion_model_amd as pkg
            dm = pkg.make_diffusion_class(self.wl["variant"])(self.model, pkg.BetaScheduler(T, device=dev).get_schedule("linear"), T)
        kw = dict(labels=self.labels) if self.wl["variant"] == 3 else {}
        if self.args.noise == "philox":
            kw["in_kernel_noise"] = True
        eng = self.model._engine
        fallbacks0 = eng.range_fallbacks
        barrier()
        t0 = time.perf_counter()
        with torch.no_grad():
            stages = dm.sample(self.shape, dev, **kw)
        barrier()
        dt = time.perf_counter() - t0
        assert len(stages) == 10 and stages[-1][1] == 0 and torch.isfinite(stages[-1][0]).all()
        res = dict(T=T, seconds=round(dt, 4), ms_per_step=round(dt / T * 1e3, 3), stages=len(stages), shard=list(self.shape),
                   range_fallbacks=eng.range_fallbacks - fallbacks0, arithmetic_at_the_end=eng.effective_math(),
                   largest_sample_magnitude=round(float(stages[-1][0].abs().max()), 1),
                   what=f"DiffusionModel.sample(x_shape, device{', labels' if kw.get('labels') is not None else ''}) over the full "
                        f"T = {T} linear schedule on every rank's shard, wall clock between two barriers, including the ten stage "
                        "copies to pinned host memory")
        if self.wl["cin"] == 1 and self.wl["hw"] == 32 and self.ctx["rank"] == 0:
            # the "+classifier.py path" of this configuration: score the samples with the MNIST classifier (random weights
            # here: the trained ones are not part of the synthetic workload) and reduce to the Inception-Score terms
            import diffusion_model_amd as pkg
            from diffusion_model_amd import scorer
            torch.manual_seed(1)
            clf = pkg.SimpleCNN().to(dev).eval()
            imgs = ((stages[-1][0] + 1) / 2).clamp(0, 1)   # the callers' post-processing (first-iteration/generate_samples.py:36-37)
            scorer.inception_score(imgs, clf)              # first call packs the weights
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            mean, _ = scorer.inception_score(imgs, clf)
            res["score"] = dict(seconds=round(time.perf_counter() - t0, 5), images=int(imgs.shape[0]), inception_score=round(float(mean), 4),
                                what="SimpleCNN logits + softmax + 10-split Inception-Score reduction of rank 0's final samples (host "
                                     "images in, scalar out), HIP kernels")
        return res, dt

    # ---- report -------------------------------------------------------------------------------
    def report(self, dt):
        args, ctx, wl = self.args, self.ctx, self.wl
        world, multi = ctx["world"], ctx["multi"]
        sec_per_step = dt / args.steps
        total_b = self.global_batch
        value = total_b / (T_FULL * sec_per_step) if self.sampling else total_b / sec_per_step
        agg = {}
        for label, flops, e0, e1 in self.prof:
            a = agg.setdefault(label, [0.0, 0.0, 0])
            a[0] += flops
            a[1] += e0.elapsed_time(e1) * 1e-3
            a[2] += 1
        if os.environ.get("DDPM_BENCH_PER_SHAPE") and ctx["rank"] == 0:   # diagnostic: the same figures per distinct launch shape
            shapes = {}
            for label, flops, e0, e1 in self.prof:
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
        cfg = {"workload": wl["name"], "per_gpu_batch": self.B, "global_batch": total_b, "conv_math": self.model._engine.effective_math(),
               "weights": "reference-default initialisation under torch.manual_seed(0), BatchNorm running statistics and attention gates "
                          "randomised (tests/golden/seeding.py)"}
        if self.sampling:
            cfg["weights"] += ("; output layer scaled by 1/64 so that this untrained network's 1000-step trajectory stays within a few "
                               "thousand, as a trained network's does, instead of growing to 4e7")
            cfg.update(T=T_FULL, step="one denoising step as DiffusionModel.sample() runs it: noise draw (%s) + UNet eval forward whose final "
                                      "1x1 conv is fused with the reverse-step update" % ("in-kernel Philox" if args.noise == "philox" else "torch.randn_like"))
        else:
            bn = "global batch statistics (all-reduce of per-rank sums, forward and backward)" if (multi and args.batchnorm == "sync") \
                else "per-rank batch statistics"
            par = "single GPU"
            if multi:
                backend = "RCCL" if args.backend == "nccl" else args.backend
                par = (f"dp{world}: {backend} all-reduce of {args.bucket_mb} MiB gradient buckets launched from inside the backward"
                       if args.allreduce == "bucketed" else f"dp{world}: {backend} all-reduce of the flattened gradients after the backward")
            cfg.update(optimizer=("torch.optim.Adam(lr=2e-4)" if args.optimizer == "torch" else
                                  "Adam(lr=2e-4), torch.optim.Adam's update as one HIP launch per 64 tensors"),
                       batchnorm=bn, parallelism=par, last_loss=round(self.state["loss"], 5))
        # memory-side bytes per launch of the dominant kernel: not measurable live (hardware counters need their own
        # rocprofv3 --pmc passes), so taken from the newest committed summary of tools/pmc_traffic.sh for this workload
        traffic, traffic_src = None, None
        kernel_symbol = {"conv3x3_halo": "conv3x3_halo_kernel", "attention_fwd": "attention_fwd_x3_kernel"}.get(dom[0].split("<")[0]) \
            if arithmetic_of(dom[0]) else None
        kind = "sample" if self.workload == "sample" else "train"
        tfile = next((f for f in (os.path.join(ROOT, "profiles", f"r{r:02d}_pmc_traffic_{kind}.txt") for r in (3, 2, 1)) if os.path.exists(f)), None)
        if kernel_symbol and self.workload in ("train", "sample") and tfile and (not self.sampling or self.B == 512 or "r03" in tfile):
            for line in open(tfile):
                f = line.split()
                if len(f) > 4 and kernel_symbol in line and f[0].isdigit():
                    traffic = int(float(f[3]) * 1e6)
                    traffic_src = ("bytes per launch, mean over this kernel's launches of one step: rocprofv3 --pmc FETCH_SIZE (x the factor "
                                   "calibrated for this kernel's access pattern on a known byte count, tools/pmc_calibrate.sh: FETCH_SIZE "
                                   "counts 64 bytes per read request of any size) + WRITE_SIZE in separate passes, profiles/" +
                                   os.path.basename(tfile))
                    break
        arith = ARITHMETIC.get(arithmetic_of(dom[0]))
        return {
            "metric": wl["metric"], "value": round(value, 4), "unit": wl["unit"],
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(sec_per_step * 1e3, 3),
            "higher_is_better": True, "scaling": "strong" if self.sampling else "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": cfg,
            "roofline": {"bound": "mfma", "kernel": dom[0], "achieved": round(achieved, 2), "peak": round(kernel_peak(dom[0]), 1),
                         "unit": "TFLOP/s", "frac": round(achieved / kernel_peak(dom[0]), 4), "traffic": traffic, "traffic_source": traffic_src,
                         "x_fp32_matrix_peak": round(achieved / FP32_MFMA_PEAK_TFLOPS, 3),   # against v_mfma_f32_32x32x2_f32's 157.3 TFLOP/s, the fp32 ceiling the split arithmetic replaces
                         "sustained_peak": (round(arith["sustained"] / arith["products"], 1) if arith else None),
                         "sustained_peak_note": (f"a loop of nothing but independent {arith['inst']} ({arith['probe']}) sustains "
                                                 f"{arith['sustained'] / 1e3:.2f} PFLOP/s on this chip (1.6-1.8 GHz under that load, not the 2.4 GHz "
                                                 "of the nominal peak)" if arith else None),
                         "arithmetic": arith["text"] if arith else "v_mfma_f32_32x32x2_f32",
                         "avg_launch_ms": round(dom[1][1] / dom[1][2] * 1e3, 4), "sampled_launches": dom[1][2],
                         "measured": f"HIP events around every {max(1, args.profile_stride)}-th MFMA launch of the timed steps (rotating over the launch sites)"
                                     if not multi else "HIP events in a 2-step pre-pass before the process group is created"},
            "kernels": kernels,
        }


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as children (torch.distributed.run on
    127.0.0.1) before this process has touched a GPU, pass rank 0's JSON line through, leave with their exit code."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    log(f"no launcher in the environment: starting {args.gpus} ranks: {' '.join(cmd)}")
    proc = subprocess.run(cmd, stdout=subprocess.PIPE, text=True)
    lines = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")]
    if lines:
        print(lines[-1], flush=True)
    for ln in proc.stdout.splitlines():
        if not ln.startswith("{"):
            print(ln, file=sys.stderr)
    sys.exit(proc.returncode if proc.returncode or lines else 1)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="both", choices=["both"] + list(WORKLOADS),
                    help="both (default) = train (the line's top level) + sample (its \"sampling\" object)")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's; sampling: its global batch / N)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-full-call", action="store_true", help="sampling: skip the real T=1000 sample() call (value = the K-step figure)")
    ap.add_argument("--no-full-range", action="store_true", help="skip the K steps in the three-plane bf16 arithmetic")
    ap.add_argument("--full-T", type=int, default=T_FULL, help="timesteps of the real sample() call (a shorter schedule for quick runs)")
    ap.add_argument("--optimizer", default="hip", choices=["hip", "torch"], help="training: diffusion_model_amd.optim.Adam or torch.optim.Adam")
    ap.add_argument("--profile-stride", type=int, default=7,
                    help="bracket every N-th MFMA launch of the timed steps with HIP events (1 = all; events cost GPU time)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse on one GPU)")
    ap.add_argument("--allreduce", default="bucketed", choices=["bucketed", "flat"],
                    help="data-parallel training: gradient buckets all-reduced while the backward runs, or one flat all-reduce after it")
    ap.add_argument("--bucket-mb", type=int, default=8, help="gradient bucket size in MiB (bucketed all-reduce)")
    ap.add_argument("--batchnorm", default="sync", choices=["local", "sync"],
                    help="data-parallel training: global batch statistics (the reference's semantics at the global batch; default), or per-rank ones")
    ap.add_argument("--noise", default="torch", choices=["torch", "philox"],
                    help="sampling: per-step noise from torch.randn_like (sample()'s default) or drawn inside the fused last kernel")
    ap.add_argument("--same-device", action="store_true", help="rehearsal: every rank uses GPU 0")
    ap.add_argument("--settle", type=float, default=1.5, help="seconds of untimed steps before the warm-up steps")
    ap.add_argument("--force-dist", action="store_true", help="rehearsal: take the multi-GPU code path (process group, collectives) "
                                                              "even with one rank")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1 and "RANK" not in os.environ:
        spawn_ranks(args)               # does not return

    import torch

    # stdout carries exactly one JSON line: native libraries that print there (RCCL's version banner at communicator
    # creation, gloo's connection notes) are sent to stderr by pointing fd 1 at fd 2; the line goes to the saved fd
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if args.same_device:
        local = 0
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    multi = world > 1 or args.force_dist
    ctx = dict(dev=dev, world=world, rank=rank, local=local, multi=multi, dist=None)
    log(f"world={world} device={dev} workload={args.workload}")
    if os.environ.get("DDPM_BENCH_TRACE"):
        import faulthandler
        faulthandler.dump_traceback_later(int(os.environ["DDPM_BENCH_TRACE"]), exit=True)
    from diffusion_model_amd import parallel

    names = ["train", "sample"] if args.workload == "both" else [args.workload]
    if args.workload == "both" and args.batch:
        sys.exit("--batch needs a single --workload")

    def barrier():
        torch.cuda.synchronize()
        dist = ctx["dist"]
        if dist is not None:
            log(f"rank {rank}: entering barrier", True)
            if args.backend == "nccl":
                dist.barrier(device_ids=[local])
            else:
                dist.barrier()
            torch.cuda.synchronize()
            log(f"rank {rank}: left barrier", True)

    def max_over_ranks(dt):
        if ctx["dist"] is None:
            return dt
        tt = torch.tensor([dt], device=dev, dtype=torch.float64)
        parallel.all_reduce_(tt, op=ctx["dist"].ReduceOp.MAX)
        return tt.item()

    results = []
    for name in names:
        # one workload at a time: the sampling half's 4096-image activations and the training half's tape never coexist
        j = Job(name, args, ctx)
        if multi:
            log(f"{name}: multi-GPU profiling pre-pass (no collectives)")
            j.prepass()
            if ctx["dist"] is None:
                import torch.distributed as dist_mod
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29533")
                if os.environ.get("DDPM_BENCH_NO_PG"):      # diagnostic: the multi-GPU control flow without a process group
                    dist_mod = None
                elif args.backend == "nccl":
                    dist_mod.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
                else:
                    dist_mod.init_process_group(args.backend, rank=rank, world_size=world)
                ctx["dist"] = dist_mod
            j.attach_distributed()
        log(f"{name}: settling")
        n = j.settle()
        log(f"{name}: {n} settling steps; warm-up and timed steps")
        dt = max_over_ranks(j.timed(barrier))
        log(f"{name}: {dt / args.steps * 1e3:.2f} ms/step")
        res = j.report(dt)
        if not args.no_full_range:
            fr = j.full_range_steps(barrier)
            if fr is not None:
                fr = max_over_ranks(fr)
                per = fr / args.steps
                res["full_range"] = dict(conv_math="bf16x3", ms_per_step=round(per * 1e3, 3),
                                         value=round(j.global_batch / (T_FULL * per) if j.sampling else j.global_batch / per, 4), unit=j.wl["unit"],
                                         what="the same K steps with every operand as three bf16 planes (six cross products, no range limit): "
                                              "the arithmetic a call falls back to when an operand leaves fp16's range")
                log(f"{name}: {per * 1e3:.2f} ms/step in the full-range arithmetic")
        if j.sampling:
            res["per_step"] = dict(value=res["value"], unit=res["unit"], ms_per_step=res["ms_per_step"], steps=args.steps,
                                   what="global_batch / (1000 * seconds per timed denoising step): the K-step figure")
            if not args.no_full_call:
                log(f"{name}: one real sample() call of T = {args.full_T} at {j.shape}")
                call_res, wall = j.full_call(barrier, args.full_T)
                wall = max_over_ranks(wall)
                call_res["seconds"] = round(wall, 4)
                call_res["timed_step_ms"] = res["ms_per_step"]
                res["full_call"] = call_res
                if args.full_T == T_FULL:
                    res["value"] = round(j.global_batch / wall, 4)     # the metric itself: images per second of the whole T-step call
                    res["value_source"] = "full_call: global_batch / wall seconds of one DiffusionModel.sample() over T = 1000"
                log(f"{name}: sample() took {wall:.2f} s = {j.global_batch / wall:.2f} images/s")
            if rank == 0 and not multi:
                res["hbm_kernel"] = reverse_step_bandwidth(dev, j.B * j.wl["cin"] * j.wl["hw"] * j.wl["hw"])
        results.append((j.workload, j.wl, j.B, res))
        del j
        torch.cuda.empty_cache()

    if rank == 0:
        out = results[0][3]
        if not multi and not args.no_cpu_baseline:
            for workload, wl, B, res in results:
                log(f"{workload}: CPU baseline")
                res["cpu_baseline"] = cpu_baseline(workload, wl, B)
        if len(results) > 1:
            samp = results[1][3]
            out["sampling"] = {k: samp[k] for k in ("metric", "value", "value_source", "unit", "scaling", "ms_per_step", "config", "per_step", "full_call",
                                                    "full_range", "roofline", "kernels", "hbm_kernel", "cpu_baseline") if k in samp}
            out.setdefault("hbm_kernel", samp.get("hbm_kernel"))
        os.write(real_stdout, (json.dumps(out) + "\n").encode())
    if ctx["dist"] is not None:
        ctx["dist"].destroy_process_group()


if __name__ == "__main__":
    main()
