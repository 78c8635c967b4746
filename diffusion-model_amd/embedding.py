"""Sinusoidal timestep embedding module (interface of first-iteration/embeddings.py:6-20)."""
import ctypes as C

import torch
import torch.nn as nn

from ._lib import call


class SinusoidalTimeEmbedding(nn.Module):
    def __init__(self, embedding_dim):
        super().__init__()
        self.embedding_dim = embedding_dim

    def forward(self, timesteps):
        if not timesteps.is_cuda:
            raise RuntimeError("SinusoidalTimeEmbedding runs on the MI355X only: move `timesteps` to the GPU")
        if timesteps.dtype not in (torch.int64, torch.float32):
            timesteps = timesteps.to(torch.float32 if timesteps.is_floating_point() else torch.int64)
        t = timesteps.contiguous()
        half = self.embedding_dim // 2
        out = torch.empty((t.shape[0], 2 * half), device=t.device, dtype=torch.float32)
        call("ddpm_sinusoidal_embedding", C.c_void_p(t.data_ptr()), int(t.dtype == torch.float32), t.shape[0],
             self.embedding_dim, C.c_void_p(out.data_ptr()), C.c_void_p(torch.cuda.current_stream().cuda_stream))
        return out
