"""The allocation-free BatchNorm statistics entry point (ddpm_bn_stats_noalloc: partial sums in a
buffer built into the library's device image) against ddpm_bn_stats (caller workspace), through the
C ABI: bit-identical mean / invstd / running statistics on two calls in a row, the second on a
smaller input (the shared buffer still holds the first call's partials), and a clear error beyond
the channel limit. Collected last: a failure here must not hide other tests behind `pytest -x`."""
import pytest
import torch

from diffusion_model_amd import _lib
from gpu_util import P, call, rnd, stream

pytestmark = pytest.mark.gpu


def _stats(entry, x, c, with_ws):
    rows, ld = x.numel() // x.shape[-1], x.stride(-2)          # a channel slice of a wider buffer keeps the buffer's pixel stride
    dev = x.device
    mean, invstd = torch.empty(c, device=dev), torch.empty(c, device=dev)
    rm, rv = torch.full((c,), 0.25, device=dev), torch.full((c,), 1.5, device=dev)
    args = [P(x), ld, rows, c, 1e-5, 0.1]
    if with_ws:
        ws = torch.empty(_lib.lib().ddpm_reduce_workspace_doubles(c), dtype=torch.float64, device=dev)
        args.append(P(ws))
    call(entry, *args, P(mean), P(invstd), P(rm), P(rv), stream())
    torch.cuda.synchronize()
    return mean.cpu(), invstd.cpu(), rm.cpu(), rv.cpu()


def test_noalloc_bn_stats_matches_workspace_variant_on_two_calls():
    limit = _lib.lib().ddpm_bn_stats_noalloc_max_channels()
    assert limit == 512
    # first call: the widest tensor the buffer allows and enough rows to use every partial block;
    # second call: fewer rows AND fewer channels, read through a wider pixel stride
    big = (rnd((64, 28, 28, 512), 1) * 2.0 + 0.5).cuda()
    small = rnd((3, 7, 7, 96), 2).cuda()[..., :64]                 # ld = 96, c = 64
    for x, c in ((big, 512), (small, 64)):
        got = _stats("ddpm_bn_stats_noalloc", x, c, with_ws=False)
        want = _stats("ddpm_bn_stats", x, c, with_ws=True)
        for g, w, name in zip(got, want, ("mean", "invstd", "running_mean", "running_var")):
            assert torch.equal(g, w), f"{name} differs at c={c}"
        ref = x[..., :c].double().reshape(-1, c)
        assert torch.allclose(got[0].double(), ref.mean(0).cpu(), atol=1e-6)
        assert torch.allclose(got[1].double(), (1.0 / torch.sqrt(ref.var(0, unbiased=False) + 1e-5)).cpu(), rtol=1e-6)


def test_noalloc_bn_stats_rejects_more_channels_than_its_buffer_holds():
    x = torch.zeros((1, 2, 2, 1024), device="cuda")
    mean, invstd = torch.empty(1024, device="cuda"), torch.empty(1024, device="cuda")
    with pytest.raises(_lib.DdpmError, match="exceed the 512"):
        call("ddpm_bn_stats_noalloc", P(x), 1024, 4, 1024, 1e-5, 0.1, P(mean), P(invstd), None, None, stream())
