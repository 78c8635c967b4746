"""fp32-grade accuracy of the split-bf16 convolution kernels, at kernel level.

The forward / data-gradient convolutions compute every fp32 product as six bf16 MFMA products and
fold the MFMA partial sum into a master accumulator every 64 (generic kernel) or 144 (halo kernel)
elements of the contraction. The worst case for that scheme is a long contraction over like-signed
data (a post-ReLU activation against positive weights: no cancellation, so a truncating accumulator
drifts). Here: K = 9*512 = 4608 on such data, error against an fp64 convolution, compared with the
error of the plain fp32-MFMA kernels (DDPM_CONV_MATH=fp32, read once at library load, hence a child
process) on the same operands. The split kernels must be within 2x of the fp32 kernels' error and
must not carry a systematic bias."""
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
x, w = torch.from_numpy(np.load({xp!r})), torch.from_numpy(np.load({wp!r}))
np.save({op!r}, T.run_conv(x, w).numpy())
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


@pytest.mark.parametrize("cout,kernel", [(128, "halo kernel, master fold every 144"), (32, "generic kernel, master fold every 64")])
@pytest.mark.parametrize("like_signed", [True, False])
def test_split_bf16_conv_error_is_fp32_grade_at_k4608(tmp_path, cout, kernel, like_signed):
    assert os.environ.get("DDPM_CONV_MATH") != "fp32", "this test compares the default kernels with the fp32 ones"
    x, w = rnd((2, 512, 16, 16), 1), rnd((cout, 512, 3, 3), 2, 0.02)
    if like_signed:
        x, w = x.abs(), w.abs()
    ref = F.conv2d(x.double(), w.double(), padding=1)
    got = run_conv(x, w)
    xp, wp_, op = (str(tmp_path / n) for n in ("x.npy", "w.npy", "o.npy"))
    np.save(xp, x.numpy())
    np.save(wp_, w.numpy())
    code = CHILD.format(root=ROOT, tests=os.path.join(ROOT, "tests"), xp=xp, wp=wp_, op=op)
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, DDPM_CONV_MATH="fp32"), capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr[-2000:]
    fp32 = torch.from_numpy(np.load(op))
    scale = ref.abs().max().item()
    e_split, e_fp32 = (got.double() - ref), (fp32.double() - ref)
    m_split, m_fp32 = e_split.abs().max().item() / scale, e_fp32.abs().max().item() / scale
    b_split = e_split.mean().item() / scale
    print(f"\n[accuracy] K=4608 cout={cout} ({kernel}) like_signed={like_signed}: max|err|/max|ref| split-bf16 {m_split:.3e}, "
          f"fp32 MFMA {m_fp32:.3e}; mean err/max split-bf16 {b_split:+.3e}, fp32 MFMA {e_fp32.mean().item() / scale:+.3e}")
    assert m_fp32 < 2e-5, "the comparator itself is off"
    assert m_split <= 2.0 * m_fp32, (m_split, m_fp32)
    assert abs(b_split) <= 1e-6, f"systematic bias {b_split:+.3e} of the tensor's scale"
