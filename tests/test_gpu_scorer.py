"""The scorer on the GPU through the drop-in SimpleCNN and the C ABI: logits against the reference's own
outputs (tests/golden/classifier.npz: the shipped mnist_classifier.pth weights on seeded images) and the CPU
oracle; Inception-Score terms, mean / covariance and Frechet distance against the oracle and the golden
values. Tolerances: logits 2e-5 of their scale (K = 4096 fp32 contraction in another order), scores 1e-5
relative, covariance 1e-9 relative to its largest entry (float64 sums on both sides)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from gpu_util import P, assert_close, call, rnd, stream
from oracle import ddpm_oracle as O
import diffusion_model_amd as pkg
from diffusion_model_amd import _lib, scorer

pytestmark = pytest.mark.gpu


def _model(g):
    sd = {k[3:]: torch.from_numpy(g[k]) for k in g.files if k.startswith("w::")}
    m = pkg.SimpleCNN()
    m.load_state_dict(sd)                                  # same keys as the reference's checkpoint
    return m.cuda().eval(), sd


def test_simple_cnn_logits_match_the_reference():
    g = load_golden("classifier.npz")
    m, sd = _model(g)
    for tag in ("gen", "real"):
        x = torch.from_numpy(g[f"{tag}_x"])
        with torch.no_grad():
            got = m(x.cuda()).cpu()
        assert got.shape == (x.shape[0], 10)
        assert_close(got, torch.from_numpy(g[f"{tag}_logits"]), 2e-5, f"{tag} logits vs reference")
        assert_close(got, O.simple_cnn_forward(sd, x), 2e-5, f"{tag} logits vs oracle")
        # chunking the batch (the reference feeds 32 images at a time) changes nothing, bit for bit
        with torch.no_grad():
            parts = torch.cat([m(x[i:i + 32].cuda()) for i in range(0, x.shape[0], 32)]).cpu()
        assert torch.equal(parts, got)
    with pytest.raises(RuntimeError, match="MI355X"):
        m(torch.zeros(1, 1, 32, 32))
    with pytest.raises(RuntimeError, match="32,32"):
        m(torch.zeros(1, 1, 28, 28).cuda())
    assert m(torch.zeros(0, 1, 32, 32).cuda()).shape == (0, 10)


def test_inception_score_and_fid_match_the_reference():
    g = load_golden("classifier.npz")
    m, _ = _model(g)
    gen, real = torch.from_numpy(g["gen_x"]), torch.from_numpy(g["real_x"])
    mean, std = scorer.inception_score(gen, m, splits=10)          # images on the host, as in the reference
    assert abs(mean.item() - g["is_mean_std"][0]) <= 1e-5 * g["is_mean_std"][0]
    assert abs(std.item() - g["is_mean_std"][1]) <= 1e-3 * g["is_mean_std"][1]
    fid = scorer.calculate_fid(real, gen, m, batch_size=64)
    assert abs(fid - g["fid"][0]) <= 1e-4 * max(1.0, abs(g["fid"][0]))


@pytest.mark.parametrize("n,k,splits", [(320, 10, 10), (37, 10, 10), (4096, 10, 10), (50, 3, 7), (10, 16, 10)])
def test_split_scores_against_the_oracle(n, k, splits):
    logits = rnd((n, k), n + k, 3.0)
    got = scorer.split_scores(logits.cuda(), splits).cpu()
    want = O.inception_split_scores(logits, splits)
    assert torch.allclose(got, want, rtol=1e-5, atol=0), (got, want)
    probs = torch.empty((n, k), device="cuda")
    ld = logits.cuda()
    call("ddpm_softmax_rows", P(ld), n, k, P(probs), stream())
    assert_close(probs.cpu(), torch.softmax(logits, 1), 1e-6, "softmax")


def test_split_scores_reject_empty_splits_and_wide_rows():
    x = torch.zeros((5, 10), device="cuda")
    out = torch.empty(10, device="cuda")
    with pytest.raises(_lib.DdpmError):
        call("ddpm_inception_split_scores", P(x), 5, 10, 10, P(out), stream())     # 5 // 10 == 0 rows per split
    with pytest.raises(_lib.DdpmError):
        call("ddpm_softmax_rows", P(torch.zeros((2, 32), device="cuda")), 2, 32, P(torch.zeros((2, 32), device="cuda")), stream())


@pytest.mark.parametrize("n,k", [(2, 10), (200, 10), (5000, 10), (70000, 4), (333, 16)])
def test_mean_and_covariance_against_numpy(n, k):
    x = rnd((n, k), n, 2.0) + torch.arange(k) * 0.5
    mean, cov = scorer.feature_statistics(x.cuda())
    want_mean, want_cov = np.mean(x.numpy().astype(np.float64), axis=0), np.cov(x.numpy(), rowvar=False)
    assert np.abs(mean - want_mean).max() <= 1e-12 * max(1.0, np.abs(want_mean).max())
    assert np.abs(cov - want_cov).max() <= 1e-9 * np.abs(want_cov).max()
