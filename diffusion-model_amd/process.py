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

    def _fusable(self):
        """The model is this package's UNet in eval mode: its last layer can be fused with the reverse step."""
        m = self.model
        return hasattr(m, "denoise_step") and hasattr(m, "_engine") and not m.training and self.fuse_final_step

    fuse_final_step = True     # class-level switch (tests compare the fused and the two-kernel paths)

    def _sample(self, x_shape, device, labels, cfg_scale, noise, in_kernel_noise=False):
        """noise: optional (x_T, [z_{T-1}, ..., z_1]) to inject instead of drawing (parity tests).
        in_kernel_noise: draw the per-step noise inside the fused step kernel from a Philox stream (seeded from
        torch's CPU generator, one stream per step) instead of torch.randn_like: no z tensor, no separate launch."""
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
        fused = self._fusable() and x_t.dim() == 4
        philox = bool(in_kernel_noise) and zs is None
        if philox and not fused:
            raise RuntimeError("in_kernel_noise needs this package's UNet in eval mode (the noise is drawn by its fused last kernel)")
        seed = int(torch.randint(0, 2 ** 62, (1,)).item()) if philox else 0
        # Range watch (fp16-pair arithmetic only): the kernels raise a device flag when an operand lies beyond fp16's
        # range, which the pair clips where fp32 does not. The flag is read at every stage hand-off; if it is up, the
        # steps since the previous hand-off are repeated from a saved copy of x_t (same noise: the generator's state is
        # restored) with three bf16 planes per operand, and the network stays there. A trained network keeps x_t at
        # unit scale and never gets here; an untrained one whose samples grow without bound does.
        eng = getattr(self.model, "_engine", None)
        watch = eng is not None and eng.range_watching()
        ckpt = None
        if watch:
            eng.range_poll()
            if eng._range_unread or eng._range_pending is not None:
                # an earlier call of this network has not had its flag read yet: settle that first
                hit = eng.range_check_sync()
                eng.range_poll()
                eng._range_unread = False
                if hit and eng.range_watching():
                    eng.range_fallback("an earlier call")
            watch = eng.range_watching()
        if watch:
            eng.range_clear()
            ckpt = self._checkpoint(x_t, T - 1, zi, len(stages))
        external_before = eng._range_external if eng is not None else False
        if eng is not None:
            eng._range_external = True       # model calls inside the loop leave the flag to the checks below
        try:
            return self._sample_loop(x_shape, x_t, T, zs, zi, stages, labels, cfg_scale, philox, seed, fused, eng, watch, ckpt)
        finally:
            if eng is not None:
                eng._range_external = external_before

    def _sample_loop(self, x_shape, x_t, T, zs, zi, stages, labels, cfg_scale, philox, seed, fused, eng, watch, ckpt):
        t = T
        while t > 0:
            t -= 1
            t_tensor = torch.full((x_shape[0],), t, dtype=torch.long, device=x_t.device)
            c1, c2, c3 = step_coefficients_host(self._betas_h[t], self._alphas_h[t], self._abar_h[t])
            z = None
            if t > 0 and not philox:
                if zs is not None:
                    z = torch.as_tensor(zs[zi], dtype=torch.float32).to(x_t.device).contiguous()
                    zi += 1
                else:
                    z = torch.randn_like(x_t)                # per-step noise from the device generator
            if fused:
                # the UNet's final 1x1 conv and the update are one kernel: eps-hat is consumed where it is produced
                # (t_tensor is torch.full: one embedding row serves the whole batch; model.feature_maps holds the LAST
                # step's maps after the call, as after the reference's loop: the steps before it need not produce them)
                step = dict(c1=c1, c2=c2, c3=c3, noise_mode=0 if t == 0 else (2 if philox else 1), z=z, seed=seed, call=t,
                            t_uniform=True, want_fms=(t == 0))
                x_t = x_t.contiguous()
                if cfg_scale > 0:
                    step.update(eps_cond=self._model_call(x_t, t_tensor, labels).contiguous(), cfg_w=float(cfg_scale))
                    x_t = self.model.denoise_step(x_t, t_tensor, None, step)
                else:
                    x_t = self.model.denoise_step(x_t, t_tensor, labels if self._passes_labels else None, step)
            else:
                eps = self._model_call(x_t, t_tensor, labels).contiguous()
                eps_u = None
                if cfg_scale > 0:
                    eps_u = self.model(x_t, t_tensor, None).contiguous()
                x_next = torch.empty_like(x_t)
                # ddpm_reverse_step blends lerp(eps_uncond, eps, w): the conditional prediction is `eps`
                call("ddpm_reverse_step", _p(x_t), _p(eps), _p(eps_u), float(cfg_scale), _p(z), c1, c2, c3, _p(x_next),
                     x_t.numel(), _stream())
                x_t = x_next
            if x_t.shape != x_shape:
                raise ValueError(f"Unexpected shape after update: {x_t.shape}, expected: {x_shape}")
            if t % (T // 10) == 0 or t == 0:
                if watch:
                    if eng.range_check_sync():
                        eng.range_fallback("sample()")
                        x_t, t, zi, keep = self._restore(ckpt)
                        del stages[keep:]
                        watch = False
                        continue
                    ckpt = self._checkpoint(x_t, t - 1, zi, len(stages) + 1)
                # stage hand-off (the reference's x_t.clone().detach().cpu()): an asynchronous copy into pinned
                # memory, ordered on the stream behind the step that produced x_t; one wait at the end instead
                # of one per stage
                host = torch.empty(x_t.shape, dtype=x_t.dtype, pin_memory=True)
                host.copy_(x_t, non_blocking=True)
                stages.append((host, t))
        self._final_on_device = x_t
        torch.cuda.current_stream(x_t.device).synchronize()
        return stages

    @staticmethod
    def _checkpoint(x_t, t_next, zi, nstages):
        """What a repeat of the steps from t_next on needs: x_t, the position in an injected noise list, the stages
        handed over so far and the state of the device generator that torch.randn_like draws the per-step noise from."""
        dev = x_t.device
        rng = torch.cuda.get_rng_state(dev) if dev.type == "cuda" else None
        return dict(x=x_t.clone(), t=t_next, zi=zi, nstages=nstages, rng=rng, dev=dev)

    @staticmethod
    def _restore(ckpt):
        if ckpt["rng"] is not None:
            torch.cuda.set_rng_state(ckpt["rng"], ckpt["dev"])
        return ckpt["x"], ckpt["t"] + 1, ckpt["zi"], ckpt["nstages"]

    def sample_images(self, x_shape, device='cpu', **kwargs):
        """One batch of finished images in [0, 1] on the host: sample() followed by the callers' `(x + 1) / 2` and
        `clamp(0, 1)` (first-iteration/generate_samples.py:34-37), applied on the GPU before the copy to the host."""
        self.sample(x_shape, device, **kwargs)
        imgs = to_images(self._final_on_device)
        self._final_on_device = None
        return imgs.cpu()


def to_images(samples):
    """clamp((x + 1) / 2, 0, 1): the two lines every caller applies to sample()'s last stage
    (first-iteration/generate_samples.py:36-37), as one kernel on a GPU tensor (same bits as the torch expression)."""
    _need_gpu(samples, "to_images")
    x = samples.contiguous().float()
    out = torch.empty_like(x)
    call("ddpm_postprocess_images", _p(x), _p(out), x.numel(), _stream())
    return out


def make_grid(images, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """torchvision.utils.make_grid for a [N, C, H, W] GPU tensor with its default value_range / scale_each
    (first-iteration/generate_samples.py:71,93 call it with nrow and normalize=True): one kernel writes the grid."""
    _need_gpu(images, "make_grid")
    if images.dim() != 4:
        raise ValueError("make_grid expects a [N, C, H, W] tensor")
    x = images.contiguous().float()
    n, c, h, w = x.shape
    lo = hi = 0.0
    if normalize:
        mn, mx = torch.aminmax(x)
        lo, hi = float(mn), float(mx)
    if n == 1:                                             # torchvision returns the (normalised, 3-channel) image itself
        y = x.expand(1, 3, h, w) if c == 1 else x
        if normalize:
            y = (y.clamp(lo, hi) - lo) / max(hi - lo, 1e-5)
        return y[0].contiguous()
    xmaps = min(nrow, n)
    ymaps = -(-n // xmaps)
    grid = torch.empty((3 if c == 1 else c, (h + padding) * ymaps + padding, (w + padding) * xmaps + padding), device=x.device)
    call("ddpm_make_grid", _p(x), n, c, h, w, int(nrow), int(padding), float(pad_value), int(bool(normalize)), lo, hi, _p(grid), _stream())
    return grid


def make_diffusion_class(iteration):
    if iteration in (1, 2):
        class DiffusionModel(_DiffusionBase):
            def sample(self, x_shape, device='cpu', *, noise=None, in_kernel_noise=False):
                """Probabilistic sampling method (DDPM-style) with x_shape handling batch size."""
                return self._sample(x_shape, device, None, 0.0, noise, in_kernel_noise)
    else:
        class DiffusionModel(_DiffusionBase):
            _passes_labels = True

            def sample(self, x_shape, device='cpu', labels=None, cfg_scale=0.0, *, noise=None, in_kernel_noise=False):
                """Probabilistic sampling method (DDPM-style)."""
                return self._sample(x_shape, device, labels, cfg_scale, noise, in_kernel_noise)
    return DiffusionModel
