"""Pins the CPU oracle against the vectors captured from the reference (tests/golden)."""
import json
import os

import numpy as np
import pytest
import torch

from conftest import GOLDEN, load_golden
from oracle import ddpm_oracle as O
from seeding import seeded_ints, seeded_state_dict, seeded_tensor

torch.set_num_threads(1)


def ulp_diff(a, b):
    ai = np.asarray(a, np.float32).view(np.int32).astype(np.int64)
    bi = np.asarray(b, np.float32).view(np.int32).astype(np.int64)
    return np.abs(ai - bi).max()


@pytest.mark.parametrize("kind", ["linear", "quadratic", "cosine"])
@pytest.mark.parametrize("T", [10, 100, 1000])
def test_tables_bit_exact(kind, T):
    g = load_golden("schedules.npz")
    betas = O.schedule_betas(kind, T)
    alphas, abar = O.schedule_tables(betas)
    assert betas.tobytes() == g[f"{kind}_{T}_betas"].tobytes()
    assert alphas.tobytes() == g[f"{kind}_{T}_alphas"].tobytes()
    assert abar.tobytes() == g[f"{kind}_{T}_alpha_bar"].tobytes()


def test_bad_schedule_name():
    with pytest.raises(ValueError):
        O.schedule_betas("sigmoid", 10)


def test_step_coefficients_within_1ulp():
    g = load_golden("schedules.npz")
    betas = O.schedule_betas("linear", 1000)
    alphas, abar = O.schedule_tables(betas)
    c = np.array([O.step_coefficients(betas, alphas, abar, t) for t in range(1000)], dtype=np.float32)
    # torch's vectorised CPU sqrt is not correctly rounded for every input: 1 ulp per op, 2 ops in c2
    assert ulp_diff(c[:, 0], g["linear_1000_c1_torchcpu"]) <= 2
    assert ulp_diff(c[:, 1], g["linear_1000_c2_torchcpu"]) <= 2
    assert ulp_diff(c[:, 2], g["linear_1000_c3_torchcpu"]) <= 1


def test_embedding():
    g = load_golden("embeddings.npz")
    e = O.time_embedding(torch.arange(1000), 128).numpy()
    np.testing.assert_allclose(e, g["E128_t0_999"], rtol=0, atol=2e-6)
    ts = torch.from_numpy(g["t_few"])
    for dim in (16, 64, 256):
        np.testing.assert_allclose(O.time_embedding(ts, dim).numpy(), g[f"E{dim}_few"], rtol=0, atol=2e-6)
    np.testing.assert_allclose(O.time_embedding(ts.float(), 128).numpy(), g["E128_float_few"], rtol=0, atol=2e-6)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_add_noise_and_step_bit_exact(tag):
    g = load_golden("elementwise.npz")
    betas = O.schedule_betas("linear", 1000)
    alphas, abar = O.schedule_tables(betas)
    noisy = O.add_noise(g[f"{tag}_x0"], g[f"{tag}_eps"], g[f"{tag}_t"], abar)
    # sqrt inside add_noise goes through torch's CPU sqrt in the reference: <= 1 ulp on the scale
    np.testing.assert_allclose(noisy, g[f"{tag}_noisy"], rtol=3e-7, atol=1e-7)
    for tt in (999, 500, 1, 0):
        c1, c2, c3 = g[f"{tag}_coef_t{tt}"]
        z = g[f"{tag}_z"] if tt > 0 else np.zeros_like(g[f"{tag}_z"])
        out = O.reverse_step(g[f"{tag}_x0"], g[f"{tag}_eps"], z, c1, c2, c3)
        assert out.tobytes() == g[f"{tag}_step_t{tt}"].tobytes()       # same coefficients -> same bits


def test_cfg_blend_bit_exact():
    g = load_golden("elementwise.npz")
    for w in (0.25, 0.5, 2.0):
        assert O.cfg_blend(g["cfg_u"], g["cfg_c"], w).tobytes() == g[f"cfg_w{w}"].tobytes()


def _tiny_sd(variant, meta):
    tmpl = {k: torch.empty(shape) for k, shape in meta[f"tiny_v{variant}"]["keys"]}
    return seeded_state_dict(tmpl, variant)


@pytest.fixture(scope="module")
def meta():
    with open(os.path.join(GOLDEN, "meta.json")) as f:
        return json.load(f)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("hw", [28, 30, 32])
def test_tiny_unet_forward(variant, hw, meta):
    g = load_golden(f"tiny_unet_v{variant}.npz")
    sd = _tiny_sd(variant, meta)
    tag = f"hw{hw}"
    x, t = torch.from_numpy(g[f"{tag}_x"]), torch.from_numpy(g[f"{tag}_t"])
    y = torch.from_numpy(g[f"{tag}_y"]) if variant == 3 else None
    with torch.no_grad():
        out, fms, _ = O.unet_forward(sd, variant, x, t, y, train=False)
    np.testing.assert_allclose(out.numpy(), g[f"{tag}_eval_out"], rtol=1e-5, atol=1e-5)
    for i, fm in enumerate(fms):
        np.testing.assert_allclose(fm.numpy(), g[f"{tag}_eval_fm{i}"], rtol=1e-5, atol=1e-5)
    if variant == 3:
        with torch.no_grad():
            o2, _, _ = O.unet_forward(sd, variant, x, t, None, train=False)
            o3, _, _ = O.unet_forward(sd, variant, x, t, y[:1], train=False)
        np.testing.assert_allclose(o2.numpy(), g[f"{tag}_eval_out_nolabel"], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(o3.numpy(), g[f"{tag}_eval_out_label1"], rtol=1e-5, atol=1e-5)
    with torch.no_grad():
        out, fms, bufs = O.unet_forward(sd, variant, x, t, y, train=True)
    np.testing.assert_allclose(out.numpy(), g[f"{tag}_train_out"], rtol=1e-4, atol=1e-5)
    for k, v in bufs.items():
        if "running_" in k:
            np.testing.assert_allclose(v.numpy(), g[f"{tag}_buf::{k}"], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("variant", [1, 2, 3])
@pytest.mark.parametrize("hw", [28, 30])
def test_tiny_unet_grads(variant, hw, meta):
    g = load_golden(f"tiny_unet_v{variant}.npz")
    sd = _tiny_sd(variant, meta)
    tag = f"hw{hw}"
    x, t = torch.from_numpy(g[f"{tag}_x"]), torch.from_numpy(g[f"{tag}_t"])
    y = torch.from_numpy(g[f"{tag}_y"]) if variant == 3 else None
    loss, grads, _, _ = O.loss_and_grads(sd, variant, x, t, torch.from_numpy(g[f"{tag}_tgt"]), y)
    np.testing.assert_allclose(loss.item(), g[f"{tag}_train_loss"][0], rtol=1e-5)
    for k, gr in grads.items():
        ref = g[f"{tag}_grad::{k}"]
        scale = max(np.abs(ref).max(), 1e-6)
        assert np.abs(gr.numpy() - ref).max() <= 2e-4 * scale + 1e-6, k


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_sampling_trajectory(variant, meta):
    g = load_golden(f"sampling_v{variant}.npz")
    sd = _tiny_sd(variant, meta)
    betas = O.schedule_betas("linear", 20)

    def model_fn(x, tt, labels):
        with torch.no_grad():
            return O.unet_forward(sd, variant, x, tt, labels, train=False)[0]

    runs = [("plain", None, 0.0)] + ([("cfg", torch.tensor([4]), 1.5)] if variant == 3 else [])
    for tag, labels, w in runs:
        rec = []
        stages = O.sample(model_fn, betas, g[f"{tag}_xT"], g[f"{tag}_noise"], labels, w, record=rec)
        assert [s[1] for s in stages] == list(g[f"{tag}_stage_t"])
        assert len(stages) == 10
        ref_calls = g[f"{tag}_eps_calls"]
        if w > 0:                       # the reference calls the model twice per step (cond, uncond)
            assert ref_calls.shape[0] == 40
        np.testing.assert_allclose(torch.stack([s[0] for s in stages]).numpy(), g[f"{tag}_stage_x"],
                                   rtol=1e-4, atol=1e-4)


def test_training_trace(meta):
    g = load_golden("training_v2.npz")
    sd = _tiny_sd(2, meta)
    betas = O.schedule_betas("linear", 1000)
    _, abar = O.schedule_tables(betas)
    params = {k: v.clone() for k, v in sd.items() if O.is_param(k)}
    state, losses = {}, []
    cur = dict(sd)
    for s in range(10):
        x0 = seeded_tensor((4, 1, 28, 28), 300, f"x0_{s}", "uniform")
        noise = seeded_tensor((4, 1, 28, 28), 300, f"n_{s}")
        t = seeded_ints((4,), 1000, 300, f"t_{s}")
        noisy = torch.from_numpy(O.add_noise(x0.numpy(), noise.numpy(), t.numpy(), abar))
        cur.update(params)
        loss, grads, _, bufs = O.loss_and_grads(cur, 2, noisy, t, noise)
        cur.update(bufs)
        O.adam_step(params, grads, state)
        losses.append(loss.item())
    np.testing.assert_allclose(losses, g["losses"], rtol=2e-3)


def test_scorer_oracle_matches_reference_logits_and_metrics():
    """SimpleCNN with the reference's shipped weights, Inception Score and Frechet distance (golden: classifier.npz)."""
    g = load_golden("classifier.npz")
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    assert sorted(sd) == ["conv1.bias", "conv1.weight", "conv2.bias", "conv2.weight", "fc1.bias", "fc1.weight", "fc2.bias", "fc2.weight"]
    for tag in ("gen", "real"):
        got = O.simple_cnn_forward(sd, torch.from_numpy(g[f"{tag}_x"]))
        want = torch.from_numpy(g[f"{tag}_logits"])
        assert (got - want).abs().max().item() <= 2e-5 * want.abs().max().item()
    mean, std = O.inception_score(torch.from_numpy(g["gen_logits"]), 10)
    assert abs(mean.item() - g["is_mean_std"][0]) <= 1e-5 * g["is_mean_std"][0]
    assert abs(std.item() - g["is_mean_std"][1]) <= 1e-4 * max(g["is_mean_std"][1], 1e-3)
    f = O.fid(g["real_logits"], g["gen_logits"])
    assert abs(f - g["fid"][0]) <= 1e-6 * max(1.0, abs(g["fid"][0]))
