"""Beta schedules and derived tables, computed on the host by the C-ABI library.

Interface of first-iteration/beta_scheduler.py:5-36 (identical in all three iterations).
"""
import ctypes as C

import numpy as np
import torch

from ._lib import call

_KINDS = {"linear": 0, "quadratic": 1, "cosine": 2}


def betas_host(kind, timesteps, beta_start, beta_end):
    out = np.empty(int(timesteps), dtype=np.float32)
    call("ddpm_schedule_betas", _KINDS[kind], int(timesteps), float(beta_start), float(beta_end),
         out.ctypes.data_as(C.c_void_p))
    return out


def tables_host(betas):
    """alphas and alpha_bar (fp32 numpy) for fp32 betas; alpha_bar accumulates in float64."""
    betas = np.ascontiguousarray(betas, dtype=np.float32)
    alphas, abar = np.empty_like(betas), np.empty_like(betas)
    call("ddpm_schedule_tables", betas.shape[0], betas.ctypes.data_as(C.c_void_p), alphas.ctypes.data_as(C.c_void_p),
         abar.ctypes.data_as(C.c_void_p))
    return alphas, abar


def step_coefficients_host(beta_t, alpha_t, alpha_bar_t):
    c = (C.c_float * 3)()
    call("ddpm_step_coefficients", float(beta_t), float(alpha_t), float(alpha_bar_t), C.cast(c, C.c_void_p))
    return float(c[0]), float(c[1]), float(c[2])


class BetaScheduler:
    def __init__(self, timesteps, beta_start=0.0001, beta_end=0.02, device='cpu'):
        self.timesteps = timesteps
        self.beta_start = beta_start
        self.beta_end = beta_end
        self.device = device

    def _make(self, kind):
        return torch.from_numpy(betas_host(kind, self.timesteps, self.beta_start, self.beta_end)).to(self.device)

    def linear(self):
        return self._make("linear")

    def quadratic(self):
        return self._make("quadratic")

    def cosine(self, s=0.008):
        if s != 0.008:
            raise NotImplementedError("the cosine schedule offset is fixed at 0.008")
        return self._make("cosine")

    def get_schedule(self, scheduler_type="linear"):
        if scheduler_type not in _KINDS:
            raise ValueError(f"Invalid scheduler type: {scheduler_type}")
        return self._make(scheduler_type)
