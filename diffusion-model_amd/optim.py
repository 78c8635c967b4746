"""`Adam`: torch.optim.Adam's update (the optimizer of the reference's training loops,
first-iteration/main.py:51,82-84; third-iteration/main.py:61) as one HIP launch per 64 tensors.

Same constructor arguments, same per-parameter state (`step`, `exp_avg`, `exp_avg_sq`) and the same
`state_dict()` layout as torch.optim.Adam, so optimizer checkpoints move both ways.  Parameters must be
contiguous fp32 tensors on the GPU; there is no CPU fallback.
"""
import ctypes as C

import torch

from ._lib import call


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, amsgrad=False, *, maximize=False):
        if not 0.0 <= lr:
            raise ValueError(f"Invalid learning rate: {lr}")
        if not 0.0 <= eps:
            raise ValueError(f"Invalid epsilon value: {eps}")
        if not 0.0 <= betas[0] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 0: {betas[0]}")
        if not 0.0 <= betas[1] < 1.0:
            raise ValueError(f"Invalid beta parameter at index 1: {betas[1]}")
        if not 0.0 <= weight_decay:
            raise ValueError(f"Invalid weight_decay value: {weight_decay}")
        if amsgrad or maximize:
            raise NotImplementedError("amsgrad / maximize are not part of the reference's optimizer and are not implemented")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay, amsgrad=False, maximize=False))

    @torch.no_grad()
    def step(self, closure=None):
        loss = None
        if closure is not None:
            with torch.enable_grad():
                loss = closure()
        for group in self.param_groups:
            by_step = {}
            for p in group["params"]:
                g = p.grad
                if g is None:
                    continue
                if g.is_sparse:
                    raise RuntimeError("Adam does not support sparse gradients")
                if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                    raise RuntimeError("Adam (HIP): parameters must be contiguous fp32 tensors on the GPU")
                if g.dtype != torch.float32 or g.device != p.device or g.shape != p.shape:
                    raise RuntimeError("Adam (HIP): gradient must match its parameter in dtype, device and shape")
                if not g.is_contiguous():
                    g = g.contiguous()
                st = self.state[p]
                if len(st) == 0:
                    st["step"] = torch.tensor(0.0, dtype=torch.float32)
                    st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                    st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["step"] += 1
                by_step.setdefault(int(st["step"].item()), []).append((p, g, st["exp_avg"], st["exp_avg_sq"]))
            for step, items in by_step.items():
                n = len(items)
                arrays = [(C.c_void_p * n)(*[t[k].data_ptr() for t in items]) for k in range(4)]
                numel = (C.c_int64 * n)(*[t[0].numel() for t in items])
                b1, b2 = group["betas"]
                call("ddpm_adam_step", n, *arrays, numel, float(group["lr"]), float(b1), float(b2), float(group["eps"]),
                     float(group["weight_decay"]), step, C.c_void_p(torch.cuda.current_stream(items[0][0].device).cuda_stream))
                # the kernel wrote the tensors behind autograd's back: record the in-place change, as any in-place
                # torch op would (saved-tensor checks and caches keyed on the version counter rely on it)
                torch.autograd.graph.increment_version([t[k] for t in items for k in (0, 2, 3)])
        return loss
