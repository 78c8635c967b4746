"""Forward noising and ancestral sampling (interface of first-iteration/diffusion.py:4-42 and
third-iteration/diffusion.py:6-53), with the per-step update as one fused HIP kernel."""
import ctypes as C

import numpy as np
import torch

from ._lib import call
from .schedule import step_coefficients_host, tables_host


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _need_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError(f"{what}: tensors must live on the MI355X (got a {t.device} tensor); there is no CPU path")


class _DiffusionBase:
    _passes_labels = False

    def __init__(self, model, beta_scheduler, num_timesteps=1000):
        self.model = model
        self.beta_scheduler = beta_scheduler
        self.num_timesteps = num_timesteps
        # Tables follow the CPU semantics of the reference bit for bit (alpha_bar accumulated in
        # float64) whatever device the betas live on; they are then placed beside the betas.
        self._betas_h = beta_scheduler.detach().to("cpu", torch.float32).numpy().copy()
        self._alphas_h, self._abar_h = tables_host(self._betas_h)
        dev = beta_scheduler.device
        self.alpha_scheduler = torch.from_numpy(self._alphas_h.copy()).to(dev)
        self.alpha_bar_scheduler = torch.from_numpy(self._abar_h.copy()).to(dev)
        self._abar_dev = {}

    def _abar_on(self, device):
        t = self._abar_dev.get(device)
        if t is None:
            t = self.alpha_bar_scheduler.to(device).contiguous()
            self._abar_dev[device] = t
        return t

    def add_noise(self, x_start, noise, t):
        _need_gpu(x_start, "add_noise")
        if x_start.requires_grad or noise.requires_grad:
            raise NotImplementedError("add_noise is not differentiable here (the training loop never needs it)")
        x0 = x_start.contiguous().float()
        nz = noise.to(x0.device).contiguous().float()
        B = x0.shape[0]
        if t.numel() != B:
            raise RuntimeError(f"add_noise: {t.numel()} timesteps for a batch of {B}")
        # Range check only where it is free (timesteps handed over on the host). Device timesteps are not read back:
        # that would be a host sync in every training step; the kernel clamps the index, so an out-of-range device
        # timestep reads a valid table entry instead of raising as the reference's CPU indexing would.
        if not t.is_cuda and t.numel() and (int(t.max()) >= len(self._abar_h) or int(t.min()) < -len(self._abar_h)):
            raise IndexError("timestep out of range of the schedule")
        tt = t.to(device=x0.device, dtype=torch.int64).contiguous()
        out = torch.empty_like(x0)
        call("ddpm_add_noise", _p(x0), _p(nz), _p(tt), _p(self._abar_on(x0.device)), len(self._abar_h), _p(out), B,
             x0.numel() // max(B, 1), _stream())
        return out

    def _model_call(self, x_t, t_tensor, labels):
        if self._passes_labels:
            return self.model(x_t, t_tensor, labels)
        return self.model(x_t, t_tensor)

    def _sample(self, x_shape, device, labels, cfg_scale, noise):
        """noise: optional (x_T, [z_{T-1}, ..., z_1]) to inject instead of drawing (parity tests)."""
        x_shape = tuple(x_shape)
        if noise is not None:
            x_t = torch.as_tensor(noise[0], dtype=torch.float32).to(device).contiguous()
            zs = noise[1]
        else:
            x_t = torch.randn(x_shape).to(device)            # x_T comes from the CPU generator
            zs = None
        _need_gpu(x_t, "sample")
        if labels is not None:
            labels = labels.to(x_t.device)
        stages, zi = [], 0
        T = self.num_timesteps
        for t in reversed(range(T)):
            t_tensor = torch.full((x_shape[0],), t, dtype=torch.long, device=x_t.device)
            eps = self._model_call(x_t, t_tensor, labels).contiguous()
            eps_u = None
            if cfg_scale > 0:
                eps_u = self.model(x_t, t_tensor, None).contiguous()
            c1, c2, c3 = step_coefficients_host(self._betas_h[t], self._alphas_h[t], self._abar_h[t])
            z = None
            if t > 0:
                if zs is not None:
                    z = torch.as_tensor(zs[zi], dtype=torch.float32).to(x_t.device).contiguous()
                    zi += 1
                else:
                    z = torch.randn_like(x_t)                # per-step noise from the device generator
            x_next = torch.empty_like(x_t)
            call("ddpm_reverse_step", _p(x_t), _p(eps), _p(eps_u), float(cfg_scale), _p(z), c1, c2, c3, _p(x_next),
                 x_t.numel(), _stream())
            x_t = x_next
            if x_t.shape != x_shape:
                raise ValueError(f"Unexpected shape after update: {x_t.shape}, expected: {x_shape}")
            if t % (T // 10) == 0 or t == 0:
                # stage hand-off (the reference's x_t.clone().detach().cpu()): an asynchronous copy into pinned
                # memory, ordered on the stream behind the step that produced x_t; one wait at the end instead
                # of one per stage
                host = torch.empty(x_t.shape, dtype=x_t.dtype, pin_memory=True)
                host.copy_(x_t, non_blocking=True)
                stages.append((host, t))
        torch.cuda.current_stream(x_t.device).synchronize()
        return stages


def make_diffusion_class(iteration):
    if iteration in (1, 2):
        class DiffusionModel(_DiffusionBase):
            def sample(self, x_shape, device='cpu', *, noise=None):
                """Probabilistic sampling method (DDPM-style) with x_shape handling batch size."""
                return self._sample(x_shape, device, None, 0.0, noise)
    else:
        class DiffusionModel(_DiffusionBase):
            _passes_labels = True

            def sample(self, x_shape, device='cpu', labels=None, cfg_scale=0.0, *, noise=None):
                """Probabilistic sampling method (DDPM-style)."""
                return self._sample(x_shape, device, labels, cfg_scale, noise)
    return DiffusionModel
