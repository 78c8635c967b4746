"""The MNIST classifier that scores generated samples, and the Inception-Score / Frechet-distance
reductions over its logits, on the MI355X.

Interfaces: `SimpleCNN` of third-iteration/classifier.py:9-24 (same parameter names, so
`mnist_classifier.pth` loads with load_state_dict), `inception_score` and `calculate_fid` of
third-iteration/metrics2.py:70-117 (there they close over a module-global classifier and device;
here the classifier is an argument). Everything up to the 10x10 matrix square root runs as HIP
kernels; `scipy.linalg.sqrtm` of that matrix stays on the host as in the reference.
"""
import ctypes as C

import numpy as np
import torch
import torch.nn as nn

from ._lib import ConvDesc, Epilogue, call, conv_math, lib, packed_floats

ACT_NONE, ACT_RELU = 0, 1


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class SimpleCNN(nn.Module):
    """conv3x3(1->32)+ReLU+pool, conv3x3(32->64)+ReLU+pool, fc 4096->128 + ReLU, fc 128->10 on 1x32x32 inputs.
    Inference only (the reference trains it with torch; the trained weights ship as mnist_classifier.pth)."""

    def __init__(self):
        super().__init__()
        self.conv1 = nn.Conv2d(1, 32, kernel_size=3, stride=1, padding=1)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=3, stride=1, padding=1)
        self.pool = nn.MaxPool2d(kernel_size=2, stride=2, padding=0)
        self.fc1 = nn.Linear(64 * 8 * 8, 128)
        self.fc2 = nn.Linear(128, 10)
        self._packed = {}

    def _apply(self, fn, *a, **k):
        out = super()._apply(fn, *a, **k)
        self.__dict__.get("_packed", {}).clear()
        return out

    def _cached(self, name, tensor, build):
        stamp = (tensor._version, tensor.data_ptr(), conv_math())     # packed weights belong to the arithmetic they were packed in
        hit = self._packed.get(name)
        if hit is None or hit[0] != stamp:
            with torch.no_grad():
                hit = (stamp, build(tensor.detach()))
            self._packed[name] = hit
        return hit[1]

    @staticmethod
    def _pack_conv3(w):
        out = torch.empty(packed_floats(9, w.shape[0], w.shape[1]), device=w.device, dtype=torch.float32)
        call("ddpm_pack_conv3x3", _p(w.contiguous()), w.shape[0], w.shape[1], 0, _p(out), _stream())
        return out

    @staticmethod
    def _pack_linear(w):
        out = torch.empty(packed_floats(1, w.shape[0], w.shape[1]), device=w.device, dtype=torch.float32)
        call("ddpm_pack_conv1x1", _p(w.contiguous()), w.shape[0], w.shape[1], 0, _p(out), _stream())
        return out

    @classmethod
    def _pack_fc1(cls, w):
        # the reference flattens NCHW ([c][y][x]); the activations here are NHWC ([y][x][c]): permute the columns once
        o, k = w.shape
        c = 64
        return cls._pack_linear(w.view(o, c, k // c).transpose(1, 2).reshape(o, k))

    def _linear(self, x2d, wp, bias, cout, act):
        n, k = x2d.shape
        out = torch.empty((n, cout), device=x2d.device, dtype=torch.float32)
        d = ConvDesc(batch=1, in_h=n, in_w=1, c1=k, ld1=k, c2=0, ld2=0, cout=cout, ldo=cout, act=act, ldr=0, out_h=n, out_w=1, off_y=0, off_x=0)
        ep = Epilogue(None, _p(bias), None, None)
        call("ddpm_conv1x1_fwd", C.byref(d), _p(x2d), None, _p(wp), C.byref(ep), _p(out), _stream())
        return out

    def forward(self, x):
        if not x.is_cuda:
            raise RuntimeError("this SimpleCNN runs on the MI355X only: move the model and its input to the GPU (there is no CPU fallback)")
        if x.dim() != 4 or x.shape[1] != 1 or x.shape[2] != 32 or x.shape[3] != 32:
            raise RuntimeError(f"SimpleCNN expects [B,1,32,32] inputs (fc1 has 64*8*8 columns), got {tuple(x.shape)}")
        x = x.contiguous().float()
        B, dev = x.shape[0], x.device
        if B == 0:
            return torch.empty((0, 10), device=dev)
        new = lambda *s: torch.empty(s, device=dev, dtype=torch.float32)   # noqa: E731
        a1 = new(B, 32, 32, 32)
        ep = Epilogue(None, _p(self.conv1.bias.detach()), None, None)
        call("ddpm_conv3x3_first_fwd", _p(x), _p(self.conv1.weight.detach().contiguous()), C.byref(ep), ACT_RELU, _p(a1), B, 1, 32, 32, 32, 32,
             _stream())
        p1 = new(B, 16, 16, 32)
        call("ddpm_pool_skip_fwd", _p(a1), 32, None, None, 0, _p(p1), 32, None, B, 32, 32, 32, _stream())
        a2 = new(B, 16, 16, 64)
        d = ConvDesc(batch=B, in_h=16, in_w=16, c1=32, ld1=32, c2=0, ld2=0, cout=64, ldo=64, act=ACT_RELU, ldr=0, out_h=16, out_w=16,
                     off_y=0, off_x=0)
        ep = Epilogue(None, _p(self.conv2.bias.detach()), None, None)
        call("ddpm_conv3x3_fwd", C.byref(d), _p(p1), None, _p(self._cached("conv2", self.conv2.weight, self._pack_conv3)), C.byref(ep),
             _p(a2), _stream())
        p2 = new(B, 8, 8, 64)
        call("ddpm_pool_skip_fwd", _p(a2), 64, None, None, 0, _p(p2), 64, None, B, 16, 16, 64, _stream())
        h = self._linear(p2.view(B, 4096), self._cached("fc1", self.fc1.weight, self._pack_fc1), self.fc1.bias.detach(), 128, ACT_RELU)
        return self._linear(h, self._cached("fc2", self.fc2.weight, self._pack_linear), self.fc2.bias.detach(), 10, ACT_NONE)


def _logits(images, classifier, batch_size):
    """classifier(images) in chunks of `batch_size` (the reference feeds 32 at a time; every image is scored on its
    own, so the chunking changes nothing but the launch count). Images may live on the host, as in the reference."""
    dev = next(classifier.parameters()).device
    outs = []
    with torch.no_grad():
        for i in range(0, images.size(0), batch_size):
            outs.append(classifier(images[i:i + batch_size].to(dev)))
    return torch.cat(outs, dim=0).contiguous()


def split_scores(logits, splits=10):
    """exp(mean KL(p(y|x) || p(y))) of each of `splits` consecutive groups of n // splits rows -> fp32 [splits] (device)."""
    n, k = logits.shape
    probs = torch.empty_like(logits)
    call("ddpm_softmax_rows", _p(logits), n, k, _p(probs), _stream())
    scores = torch.empty(splits, device=logits.device, dtype=torch.float32)
    call("ddpm_inception_split_scores", _p(probs), n, k, splits, _p(scores), _stream())
    return scores


def inception_score(images, classifier, batch_size=4096, splits=10, num_splits=None):
    """(mean, std) of the per-split scores, as torch scalars (third-iteration/metrics2.py:70-89)."""
    scores = split_scores(_logits(images, classifier, batch_size), num_splits or splits).cpu()
    return torch.mean(scores), torch.std(scores)


def feature_statistics(features):
    """float64 mean [k] and unbiased covariance [k, k] of device features [n, k], as numpy arrays."""
    n, k = features.shape
    dev = features.device
    ws = torch.empty(int(lib().ddpm_mean_cov_workspace_doubles(k)), dtype=torch.float64, device=dev)
    mean = torch.empty(k, dtype=torch.float64, device=dev)
    cov = torch.empty((k, k), dtype=torch.float64, device=dev)
    call("ddpm_mean_cov", _p(features.contiguous()), n, k, _p(ws), _p(mean), _p(cov), _stream())
    return mean.cpu().numpy(), cov.cpu().numpy()


def fid_from_statistics(mu_real, sigma_real, mu_gen, sigma_gen):
    from scipy.linalg import sqrtm
    diff = mu_real - mu_gen
    covmean = sqrtm(sigma_real @ sigma_gen)
    if isinstance(covmean, tuple):
        covmean = covmean[0]
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(np.sum(diff ** 2) + np.trace(sigma_real + sigma_gen - 2 * covmean))


def calculate_fid(real_images, generated_images, classifier, batch_size=4096):
    """Frechet distance between the classifier's logits of the two image sets (third-iteration/metrics2.py:92-117)."""
    mu_r, s_r = feature_statistics(_logits(real_images, classifier, batch_size))
    mu_g, s_g = feature_statistics(_logits(generated_images, classifier, batch_size))
    return fid_from_statistics(mu_r, s_r, mu_g, s_g)
