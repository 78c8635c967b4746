"""Helpers shared by the GPU parity tests: raw C-ABI calls on torch CUDA tensors."""
import ctypes as C

import torch

from diffusion_model_amd._lib import ConvDesc, Epilogue, call, packed_floats  # noqa: F401


_keepalive = []


def P(t):
    """Device pointer of `t`. The tensor is kept alive until the end of the test (release_kept),
    so temporaries like P(x.cuda()) cannot be recycled by the allocator before the kernel runs."""
    if t is None:
        return None
    _keepalive.append(t)
    return C.c_void_p(t.data_ptr())


def release_kept():
    if _keepalive and torch.cuda.is_available():
        torch.cuda.synchronize()
    _keepalive.clear()


def stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def to_nhwc(t):
    """[B,C,H,W] CPU tensor -> contiguous [B,H,W,C] CUDA tensor."""
    return t.permute(0, 2, 3, 1).contiguous().cuda()


def from_nhwc(t):
    return t.cpu().permute(0, 3, 1, 2).contiguous()


def rnd(shape, seed, scale=1.0):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(shape, generator=g) * scale


def assert_close(got, want, tol, what=""):
    """max |got - want| <= tol * max(1, max|want|): a tolerance relative to the tensor's scale."""
    got, want = got.double(), want.double()
    assert got.shape == want.shape, (what, got.shape, want.shape)
    err = (got - want).abs().max().item()
    ref = max(1.0, want.abs().max().item())
    assert err <= tol * ref, f"{what}: max abs err {err:.3e} > {tol:.1e} * {ref:.3e}"
