#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the reference.

Run only in the build container (needs /root/reference):
    python tests/golden/make_golden.py

The reference's Python files are imported unmodified (empty stub modules stand in
for the uninstalled plotting/dataset packages `seaborn` and `torchvision`, which
only its visualisation helpers and __main__ demos use). Only DATA is written:
inputs and the outputs the reference produced for them. Nothing of the reference's
source travels.
"""
import hashlib
import importlib
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
from seeding import seeded_ints, seeded_state_dict, seeded_tensor  # noqa: E402

REF = "/root/reference"
ITER_DIRS = {1: "first-iteration", 2: "second-iteration", 3: "third-iteration"}
BARE = ("unet", "diffusion", "beta_scheduler", "embeddings", "utils")


def _stub_missing():
    import matplotlib
    matplotlib.use("Agg")
    for name in ("seaborn", "torchvision", "torchvision.datasets", "torchvision.transforms",
                 "torchvision.utils", "torchvision.models"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules["torchvision"].datasets = sys.modules["torchvision.datasets"]
    sys.modules["torchvision"].transforms = sys.modules["torchvision.transforms"]


def load_iteration(n):
    """Import the four hot-path modules of iteration n under their bare names."""
    for m in BARE:
        sys.modules.pop(m, None)
    d = os.path.join(REF, ITER_DIRS[n])
    sys.path.insert(0, d)
    try:
        mods = {m: importlib.import_module(m) for m in ("beta_scheduler", "embeddings", "unet", "diffusion")}
    finally:
        sys.path.remove(d)
    for m in BARE:
        sys.modules.pop(m, None)
    return mods


def npf(t):
    return t.detach().cpu().numpy().copy()


def sha16(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()[:16]


def save(name, **arrays):
    path = os.path.join(HERE, name)
    np.savez(path, **arrays)
    print(f"  wrote {name}: {os.path.getsize(path)/1e6:.2f} MB, {len(arrays)} arrays")


# ----------------------------------------------------------------------------------------------
def gen_schedules(m1):
    out, meta = {}, {}
    for T in (10, 100, 1000):
        for kind in ("linear", "quadratic", "cosine"):
            betas = m1["beta_scheduler"].BetaScheduler(T).get_schedule(kind)
            dm = m1["diffusion"].DiffusionModel(None, betas, T)
            for nm, t in (("betas", betas), ("alphas", dm.alpha_scheduler), ("alpha_bar", dm.alpha_bar_scheduler)):
                a = npf(t)
                assert a.dtype == np.float32
                out[f"{kind}_{T}_{nm}"] = a
                meta[f"{kind}_{T}_{nm}"] = sha16(a)
    # per-step coefficients exactly as the sampling loop forms them (0-dim fp32 torch ops on CPU)
    betas = m1["beta_scheduler"].BetaScheduler(1000).get_schedule("linear")
    dm = m1["diffusion"].DiffusionModel(None, betas, 1000)
    b, a, ab = dm.beta_scheduler, dm.alpha_scheduler, dm.alpha_bar_scheduler
    c1 = torch.stack([1 / torch.sqrt(a[t]) for t in range(1000)])
    c2 = torch.stack([(1 - a[t]) / torch.sqrt(1 - ab[t]) for t in range(1000)])
    c3 = torch.stack([torch.sqrt(b[t]) for t in range(1000)])
    out["linear_1000_c1_torchcpu"], out["linear_1000_c2_torchcpu"], out["linear_1000_c3_torchcpu"] = npf(c1), npf(c2), npf(c3)
    save("schedules.npz", **out)
    return meta


def gen_embeddings(m1):
    out = {}
    E = m1["embeddings"].SinusoidalTimeEmbedding
    out["E128_t0_999"] = npf(E(128)(torch.arange(1000)))            # int64 timesteps, as the scripts pass
    ts = torch.tensor([0, 1, 500, 999])
    out["t_few"] = npf(ts)
    for dim in (16, 64, 256):
        out[f"E{dim}_few"] = npf(E(dim)(ts))
    out["E128_float_few"] = npf(E(128)(ts.float()))
    save("embeddings.npz", **out)
    return {"E128_t0_999": sha16(out["E128_t0_999"])}


def gen_elementwise(m1):
    """add_noise and one reverse step on seeded tensors (reference formulas run on CPU)."""
    out = {}
    betas = m1["beta_scheduler"].BetaScheduler(1000).get_schedule("linear")
    dm = m1["diffusion"].DiffusionModel(None, betas, 1000)
    for tag, shape in (("a", (4, 1, 28, 28)), ("b", (2, 3, 32, 32))):
        x0 = seeded_tensor(shape, 11, f"x0{tag}", "uniform")
        eps = seeded_tensor(shape, 11, f"eps{tag}")
        z = seeded_tensor(shape, 11, f"z{tag}")
        t = torch.tensor([999, 500, 1, 0][: shape[0]])
        out[f"{tag}_x0"], out[f"{tag}_eps"], out[f"{tag}_z"], out[f"{tag}_t"] = npf(x0), npf(eps), npf(z), npf(t)
        out[f"{tag}_noisy"] = npf(dm.add_noise(x0, eps, t))
        # reverse step through the reference's own sample(): T-step loops are too long, so drive a
        # single step by a model stub returning eps and a 1-entry schedule slice is NOT possible
        # (tables are indexed by t); instead evaluate the reference expression via sample() on a
        # short schedule below (gen_sampling). Here: the expression with the reference's tensors.
        for tt in (999, 500, 1, 0):
            beta_t, alpha_t, alpha_bar_t = dm.beta_scheduler[tt], dm.alpha_scheduler[tt], dm.alpha_bar_scheduler[tt]
            noise = z if tt > 0 else torch.zeros_like(z)
            xn = (1 / torch.sqrt(alpha_t)) * (x0 - ((1 - alpha_t) / torch.sqrt(1 - alpha_bar_t)) * eps) + torch.sqrt(beta_t) * noise
            out[f"{tag}_step_t{tt}"] = npf(xn)
            out[f"{tag}_coef_t{tt}"] = np.array([float(1 / torch.sqrt(alpha_t)),
                                                 float((1 - alpha_t) / torch.sqrt(1 - alpha_bar_t)),
                                                 float(torch.sqrt(beta_t))], dtype=np.float32)
    # classifier-free-guidance blend (third-iteration/diffusion.py:31)
    u = seeded_tensor((2, 3, 8, 8), 12, "cfg_u")
    c = seeded_tensor((2, 3, 8, 8), 12, "cfg_c")
    out["cfg_u"], out["cfg_c"] = npf(u), npf(c)
    for w in (0.25, 0.5, 2.0):
        out[f"cfg_w{w}"] = npf(torch.lerp(u, c, w))
    save("elementwise.npz", **out)


def build_model(mods, variant, base, emb, cin, num_classes=None, seed=0):
    U = mods["unet"].UNet
    if variant == 3:
        model = U(cin, cin, base, emb, num_classes=num_classes)
    else:
        model = U(cin, cin, base, emb)
    sd = seeded_state_dict(model.state_dict(), seed)
    model.load_state_dict(sd)
    return model, sd


def gen_tiny_unet(mods, variant):
    """Tiny UNet (base 8, emb 16): eval + train forward, feature maps, grads, BN buffers."""
    out, meta = {}, {}
    cin = 1 if variant < 3 else 3
    ncls = 10 if variant == 3 else None
    model, sd = build_model(mods, variant, 8, 16, cin, ncls, seed=variant)
    meta["keys"] = [[k, list(v.shape)] for k, v in sd.items()]
    for hw, B, want_grads in ((28, 2, True), (30, 2, True), (32, 2, False)):
        tag = f"hw{hw}"
        x = seeded_tensor((B, cin, hw, hw), 100 + variant, f"x{tag}")
        t = seeded_ints((B,), 1000, 100 + variant, f"t{tag}")
        tgt = seeded_tensor((B, cin, hw, hw), 100 + variant, f"tgt{tag}")
        out[f"{tag}_x"], out[f"{tag}_t"], out[f"{tag}_tgt"] = npf(x), npf(t), npf(tgt)
        args = [x, t]
        if variant == 3:
            y = seeded_ints((B,), 10, 100 + variant, f"y{tag}")
            out[f"{tag}_y"] = npf(y)
            args.append(y)
        # eval
        model.load_state_dict(sd)
        model.eval()
        with torch.no_grad():
            e = model(*args)
            out[f"{tag}_eval_out"] = npf(e)
            for i, fm in enumerate(model.feature_maps):
                out[f"{tag}_eval_fm{i}"] = npf(fm)
            if variant == 3:
                out[f"{tag}_eval_out_nolabel"] = npf(model(x, t))
                out[f"{tag}_eval_out_label1"] = npf(model(x, t, y[:1]))       # shape-[1] label broadcast
        # train (batch statistics, running stats update) + grads of an MSE loss
        model.load_state_dict(sd)
        model.train()
        model.zero_grad()
        e = model(*args)
        out[f"{tag}_train_out"] = npf(e)
        if hw == 28:
            for i, fm in enumerate(model.feature_maps):
                out[f"{tag}_train_fm{i}"] = npf(fm)
        loss = torch.nn.MSELoss()(e, tgt)
        out[f"{tag}_train_loss"] = np.array([loss.item()], dtype=np.float32)
        if want_grads:
            loss.backward()
            for k, p in model.named_parameters():
                g = p.grad if p.grad is not None else torch.zeros_like(p)
                out[f"{tag}_grad::{k}"] = npf(g)
        for k, v in model.state_dict().items():
            if "running_" in k:
                out[f"{tag}_buf::{k}"] = npf(v)
    save(f"tiny_unet_v{variant}.npz", **out)
    return meta


def gen_full_width(mods, variant):
    """Full-width (base 64, emb 128) eval eps-hat for 2 samples; weights regenerate from the seed."""
    out = {}
    cin = 1 if variant < 3 else 3
    ncls = 10 if variant == 3 else None
    model, sd = build_model(mods, variant, 64, 128, cin, ncls, seed=40 + variant)
    h = hashlib.sha256()
    for k, v in sd.items():
        h.update(k.encode())
        h.update(np.ascontiguousarray(npf(v)).tobytes())
    model.eval()
    for hw in (28, 32):
        x = seeded_tensor((2, cin, hw, hw), 200 + variant, f"x{hw}")
        t = torch.tensor([937, 12])
        out[f"hw{hw}_x"], out[f"hw{hw}_t"] = npf(x), npf(t)
        with torch.no_grad():
            if variant == 3:
                y = torch.tensor([3, 7])
                out[f"hw{hw}_y"] = npf(y)
                out[f"hw{hw}_out"] = npf(model(x, t, y))
            else:
                out[f"hw{hw}_out"] = npf(model(x, t))
            out[f"hw{hw}_fm3"] = npf(model.feature_maps[3])
    save(f"full_unet_v{variant}.npz", **out)
    return {"state_dict_sha256": h.hexdigest(), "n_params": int(sum(p.numel() for p in model.parameters()))}


class _Recorder:
    """Callable wrapper recording every eps-hat the sampling loop receives."""

    def __init__(self, model):
        self.model, self.calls = model, []

    def __call__(self, *a):
        y = self.model(*a)
        self.calls.append(y.detach().clone())
        return y


def gen_sampling(mods, variant):
    """T=20 ancestral-sampling trajectory with the tiny UNet; noise is reproducible from the seed."""
    out = {}
    cin = 1 if variant < 3 else 3
    ncls = 10 if variant == 3 else None
    T, shape = 20, (2, cin, 28, 28)
    model, sd = build_model(mods, variant, 8, 16, cin, ncls, seed=variant)
    model.eval()
    betas = mods["beta_scheduler"].BetaScheduler(T).get_schedule("linear")
    runs = [("plain", {})]
    if variant == 3:
        runs = [("plain", {}), ("cfg", {"labels": torch.tensor([4]), "cfg_scale": 1.5})]
    for tag, kw in runs:
        rec = _Recorder(model)
        dm = mods["diffusion"].DiffusionModel(rec, betas, T)
        torch.manual_seed(777)
        with torch.no_grad():
            stages = dm.sample(shape, "cpu", **kw)
        # the same draws, in the same order, from the same seed (x_T, then one per step with t>0)
        torch.manual_seed(777)
        xT = torch.randn(shape)
        zs = [torch.randn(shape) for _ in range(T - 1)]
        out[f"{tag}_xT"] = npf(xT)
        out[f"{tag}_noise"] = npf(torch.stack(zs))
        out[f"{tag}_eps_calls"] = npf(torch.stack(rec.calls))
        out[f"{tag}_stage_t"] = np.array([s[1] for s in stages], dtype=np.int64)
        out[f"{tag}_stage_x"] = npf(torch.stack([s[0] for s in stages]))
    save(f"sampling_v{variant}.npz", **out)


def gen_training(mods, variant=2):
    """10 Adam steps of the training-loop body with injected x0 / noise / t."""
    out = {}
    B, hw, steps, T = 4, 28, 10, 1000
    model, sd = build_model(mods, variant, 8, 16, 1, None, seed=variant)
    betas = mods["beta_scheduler"].BetaScheduler(T).get_schedule("linear")
    dm = mods["diffusion"].DiffusionModel(model, betas, T)
    opt = torch.optim.Adam(model.parameters(), lr=2e-4)
    crit = torch.nn.MSELoss()
    model.train()
    losses = []
    for s in range(steps):
        x0 = seeded_tensor((B, 1, hw, hw), 300, f"x0_{s}", "uniform")
        noise = seeded_tensor((B, 1, hw, hw), 300, f"n_{s}")
        t = seeded_ints((B,), T, 300, f"t_{s}")
        noisy = dm.add_noise(x0, noise, t)
        pred = model(noisy, t)
        loss = crit(pred, noise)
        opt.zero_grad()
        loss.backward()
        opt.step()
        losses.append(loss.item())
    out["losses"] = np.array(losses, dtype=np.float64)
    for k, v in model.state_dict().items():
        out[f"final::{k}"] = npf(v)
    save(f"training_v{variant}.npz", **out)


def gen_classifier():
    """The scorer: SimpleCNN (third-iteration/classifier.py) with the weights the reference ships
    (mnist_classifier.pth, loaded weights-only), logits of seeded 1x32x32 images, and the Inception Score / Frechet
    distance the reference's helper functions (third-iteration/utils.py:194-236, the importable twins of the
    functions inside the metrics2.py script) compute from them. The weights are stored as data: the GPU box has
    no reference checkout."""
    for m in ("classifier", "utils"):
        sys.modules.pop(m, None)
    d = os.path.join(REF, ITER_DIRS[3])
    sys.path.insert(0, d)
    try:
        classifier = importlib.import_module("classifier")
        utils = importlib.import_module("utils")
    finally:
        sys.path.remove(d)
    model = classifier.SimpleCNN()
    sd = torch.load(os.path.join(REF, "mnist_classifier.pth"), map_location="cpu", weights_only=True)
    model.load_state_dict(sd)
    model.eval()
    out = {f"w::{k}": npf(v) for k, v in sd.items()}
    # "generated": blurred blobs in [-1, 1] so that the classifier is not saturated; "real": another draw
    def images(n, seed):
        base = seeded_tensor((n, 1, 32, 32), seed, "img")
        sm = torch.nn.functional.avg_pool2d(base, 5, 1, 2) * 3.0
        return torch.tanh(sm + 0.5 * seeded_tensor((n, 1, 1, 1), seed, "bias"))
    gen, real = images(320, 500), images(200, 501)
    with torch.no_grad():
        lg, lr = model(gen), model(real)
    out["gen_x"], out["real_x"], out["gen_logits"], out["real_logits"] = npf(gen), npf(real), npf(lg), npf(lr)
    is_mean, is_std = utils.inception_score(gen, model, num_splits=10)
    out["is_mean_std"] = np.array([is_mean, is_std], dtype=np.float64)
    out["fid"] = np.array([float(np.real(utils.calculate_fid(lr.numpy(), lg.numpy())))], dtype=np.float64)
    for m in ("classifier", "utils"):
        sys.modules.pop(m, None)
    save("classifier.npz", **out)
    return {"weights_sha16": {k: sha16(npf(v)) for k, v in sd.items()}}


def main():
    assert os.path.isdir(REF), "reference not mounted"
    _stub_missing()
    torch.set_num_threads(1)          # deterministic reduction order on CPU
    if "--only-classifier" in sys.argv:      # adds classifier.npz without regenerating the other fixtures
        meta = json.load(open(os.path.join(HERE, "meta.json")))
        print("classifier"); meta["classifier"] = gen_classifier()
        with open(os.path.join(HERE, "meta.json"), "w") as f:
            json.dump(meta, f, indent=1)
        return
    meta = {"torch": torch.__version__, "numpy": np.__version__}
    m1, m2, m3 = load_iteration(1), load_iteration(2), load_iteration(3)
    print("schedules"); meta["schedules_sha16"] = gen_schedules(m1)
    print("embeddings"); meta["embeddings_sha16"] = gen_embeddings(m1)
    print("elementwise"); gen_elementwise(m1)
    for v, m in ((1, m1), (2, m2), (3, m3)):
        print(f"tiny unet v{v}"); meta[f"tiny_v{v}"] = gen_tiny_unet(m, v)
        print(f"full unet v{v}"); meta[f"full_v{v}"] = gen_full_width(m, v)
        print(f"sampling v{v}"); gen_sampling(m, v)
    print("training"); gen_training(m2, 2)
    print("classifier"); meta["classifier"] = gen_classifier()
    with open(os.path.join(HERE, "meta.json"), "w") as f:
        json.dump(meta, f, indent=1)


if __name__ == "__main__":
    main()
