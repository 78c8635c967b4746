"""fp32-grade accuracy of the split-arithmetic convolution kernels, at kernel level.

The forward / data-gradient convolutions compute every fp32 product on the 16-bit matrix pipe: as three cross
products of fp16 pairs (hi + remainder scaled by 2^11; the default), or as six cross products of three bf16 planes with
the MFMA partial sum folded into a master accumulator every 64 (generic kernel) or 144 (halo kernel) elements of
the contraction. The worst case for both is a long contraction over like-signed data (a post-ReLU activation
against positive weights: no cancellation, so a truncating accumulator drifts, and dropped cross terms add up).
Here: K = 9*512 = 4608 on such data and on mixed-sign data, error against an fp64 convolution, for all three
arithmetics (DDPM_CONV_MATH is read once at library load, hence child processes) on the same operands. Either split
arithmetic must be at least as accurate as the plain fp32-MFMA kernels and must not carry a systematic bias; the
fp16 pair must also hold for operands a thousand times smaller and larger (fp16 has five exponent bits)."""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import ROOT
from gpu_util import ConvDesc, Epilogue, P, call, from_nhwc, packed_floats, rnd, stream, to_nhwc

pytestmark = pytest.mark.gpu

CHILD = """
import sys, numpy as np, torch
sys.path[:0] = [{root!r}, {tests!r}]
import test_gpu_accuracy as T
cases = np.load({inp!r})
names = sorted(set(k[2:] for k in cases.files))
np.savez({out!r}, **{{n: T.run_conv(torch.from_numpy(cases["x_" + n]), torch.from_numpy(cases["w_" + n])).numpy() for n in names}})
"""


def run_conv(x, w):
    B, ci, H, W = x.shape
    co = w.shape[0]
    wp = torch.empty(packed_floats(9, co, ci), device="cuda")
    call("ddpm_pack_conv3x3", P(w.cuda().contiguous()), co, ci, 0, P(wp), stream())
    out = torch.full((B, H, W, co), float("nan"), device="cuda")
    d = ConvDesc(batch=B, in_h=H, in_w=W, c1=ci, ld1=ci, c2=0, ld2=0, cout=co, ldo=co, act=0, ldr=0, out_h=H, out_w=W)
    ep = Epilogue(None, None, None, None)
    call("ddpm_conv3x3_fwd", C.byref(d), P(to_nhwc(x)), None, P(wp), C.byref(ep), P(out), stream())
    torch.cuda.synchronize()
    return from_nhwc(out)


def _cases():
    out = {}
    for cout, kernel in ((128, "halo"), (32, "generic")):
        for like_signed in (True, False):
            for mag in ((1.0, 2.0 ** -10, 2.0 ** 10) if (cout == 128 and not like_signed) else (1.0,)):
                x, w = rnd((2, 512, 16, 16), 1) * mag, rnd((cout, 512, 3, 3), 2, 0.02)
                if like_signed:
                    x, w = x.abs(), w.abs()
                out[f"k4608_{kernel}_{'pos' if like_signed else 'mixed'}_{mag:g}"] = (x, w)
    # K = 9 * 16 = 144: no long sum hides the operands' own representation error (22-23 bits for the fp16 pair)
    out["k144"] = (rnd((4, 16, 16, 16), 3), rnd((64, 16, 3, 3), 4, 0.1))
    return out


@pytest.fixture(scope="module")
def errors(tmp_path_factory):
    """{case: {arithmetic: (max |err| / max |ref|, mean err / max |ref|)}}: every case through the three arithmetics, one child process each."""
    tmp = tmp_path_factory.mktemp("accuracy")
    cases = _cases()
    inp = str(tmp / "cases.npz")
    np.savez(inp, **{f"x_{n}": x.numpy() for n, (x, w) in cases.items()}, **{f"w_{n}": w.numpy() for n, (x, w) in cases.items()})
    refs = {n: F.conv2d(x.double(), w.double(), padding=1) for n, (x, w) in cases.items()}
    res = {n: {} for n in cases}
    for math in ("fp32", "bf16x3", "f16x2"):
        out = str(tmp / f"out_{math}.npz")
        code = CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), inp=inp, out=out)
        r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DDPM_CONV_MATH=math), capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        got = np.load(out)
        for n, ref in refs.items():
            e = torch.from_numpy(got[n]).double() - ref
            scale = ref.abs().max().item()
            res[n][math] = (e.abs().max().item() / scale, e.mean().item() / scale)
    for n, v in res.items():
        print(f"\n[accuracy] {n}: max|err|/max|ref| " + ", ".join(f"{m} {a:.3e}" for m, (a, b) in v.items())
              + "; mean err/max " + ", ".join(f"{m} {b:+.3e}" for m, (a, b) in v.items()))
    return res


def test_split_conv_error_is_fp32_grade_at_k4608(errors):
    for n, err in errors.items():
        if not n.startswith("k4608"):
            continue
        assert err["fp32"][0] < 2e-5, "the comparator itself is off"
        for math in ("bf16x3", "f16x2"):
            assert err[math][0] <= err["fp32"][0], (n, math, err)
            assert abs(err[math][1]) <= 2e-7, f"{n} {math}: systematic bias {err[math][1]:+.3e} of the tensor's scale"


def test_short_contraction_is_fp32_grade(errors):
    err = errors["k144"]
    assert err["f16x2"][0] <= 1.5 * err["fp32"][0] and err["bf16x3"][0] <= 1.5 * err["fp32"][0], err
