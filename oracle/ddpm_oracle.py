"""CPU oracle for the DDPM hot path -- TEST INFRASTRUCTURE ONLY.

A plain numpy / torch-CPU restatement of what the reference computes on this path, written
from the reference's behaviour (file:line citations are relative to the reference checkout).
It is the checker for the HIP kernels: only tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py may import it. The shipped package never does.

Pinned by: tests/golden/*.npz, produced by importing the reference itself
(tests/golden/make_golden.py); tests/test_oracle_golden.py checks every function here
against those vectors (bit-exact for tables / elementwise steps, fp32 tolerance elsewhere).
"""
import math

import numpy as np
import torch
import torch.nn.functional as F

f32 = np.float32


# ------------------------------------------------------------------------------------------
# Schedule tables -- first-iteration/beta_scheduler.py:12-36, first-iteration/diffusion.py:9-10
# ------------------------------------------------------------------------------------------
def schedule_betas(kind, timesteps, beta_start=1e-4, beta_end=0.02):
    """float64 schedule rounded once to fp32 (beta_scheduler.py:12-26)."""
    T = int(timesteps)
    if kind == "linear":
        b = np.linspace(beta_start, beta_end, T)
    elif kind == "quadratic":
        b = np.linspace(beta_start ** 0.5, beta_end ** 0.5, T) ** 2
    elif kind == "cosine":
        s = 0.008
        steps = np.arange(T + 1)
        ac = np.cos(((steps / T) + s) / (1 + s) * np.pi * 0.5) ** 2
        ac = ac / ac[0]
        b = np.clip(1 - (ac[1:] / ac[:-1]), 0, 0.999)
    else:
        raise ValueError(f"Invalid scheduler type: {kind}")          # beta_scheduler.py:36
    return b.astype(f32)


def schedule_tables(betas):
    """alphas = 1 - betas in fp32; alpha_bar = running product accumulated in float64 and
    rounded to fp32 per element, which is what torch.cumprod does for fp32 on CPU
    (diffusion.py:9-10)."""
    betas = np.asarray(betas, dtype=f32)
    alphas = (f32(1.0) - betas).astype(f32)
    acc = np.float64(1.0)
    abar = np.empty_like(alphas)
    for i, a in enumerate(alphas):
        acc = acc * np.float64(a)
        abar[i] = f32(acc)
    return alphas, abar


def step_coefficients(betas, alphas, alpha_bar, t):
    """c1 = 1/sqrt(alpha_t), c2 = (1-alpha_t)/sqrt(1-alpha_bar_t), c3 = sqrt(beta_t), each op
    rounded to fp32 with IEEE-correct sqrt/div (diffusion.py:35-37). torch's CPU sqrt is
    1 ulp off for <1% of inputs, so these match the reference within 1 ulp, not bitwise."""
    a, ab, b = f32(alphas[t]), f32(alpha_bar[t]), f32(betas[t])
    c1 = f32(1.0) / np.sqrt(a, dtype=f32)
    c2 = (f32(1.0) - a) / np.sqrt(f32(1.0) - ab, dtype=f32)
    c3 = np.sqrt(b, dtype=f32)
    return f32(c1), f32(c2), f32(c3)


# ------------------------------------------------------------------------------------------
# Elementwise process steps -- first-iteration/diffusion.py:12-15,35-37; third-iteration/diffusion.py:31
# ------------------------------------------------------------------------------------------
def add_noise(x0, noise, t, alpha_bar):
    """sqrt(abar[t_b])*x0 + sqrt(1-abar[t_b])*noise, every op separately rounded to fp32."""
    x0, noise = np.asarray(x0, f32), np.asarray(noise, f32)
    ab = np.asarray(alpha_bar, f32)[np.asarray(t, np.int64)].reshape(-1, 1, 1, 1)
    s1 = np.sqrt(ab, dtype=f32)
    s2 = np.sqrt((f32(1.0) - ab).astype(f32), dtype=f32)
    return ((s1 * x0).astype(f32) + (s2 * noise).astype(f32)).astype(f32)


def reverse_step(x, eps, z, c1, c2, c3):
    """c1*(x - c2*eps) + c3*z with five separately rounded fp32 ops (no fused multiply-add)."""
    x, eps, z = np.asarray(x, f32), np.asarray(eps, f32), np.asarray(z, f32)
    inner = (x - (f32(c2) * eps).astype(f32)).astype(f32)
    return ((f32(c1) * inner).astype(f32) + (f32(c3) * z).astype(f32)).astype(f32)


def cfg_blend(uncond, cond, w):
    """torch.lerp(uncond, cond, w): u + w*(c-u) for |w| < 0.5, else c - (c-u)*(1-w)."""
    u, c, w = np.asarray(uncond, f32), np.asarray(cond, f32), f32(w)
    d = (c - u).astype(f32)
    if abs(float(w)) < 0.5:
        return (u + (w * d).astype(f32)).astype(f32)
    return (c - (d * f32(f32(1.0) - w)).astype(f32)).astype(f32)


# ------------------------------------------------------------------------------------------
# Sinusoidal embedding -- first-iteration/embeddings.py:11-20
# ------------------------------------------------------------------------------------------
def time_embedding(t, dim):
    """[sin(t*s_k) | cos(t*s_k)], s_k = exp(-k*ln(1e4)/(half-1)) in fp32. torch tensor in/out."""
    t = torch.as_tensor(t)
    half = dim // 2
    k = torch.arange(0, half, dtype=torch.float32)
    # the reference takes the log in fp32 and divides it by (half-1) before scaling k
    scale = torch.exp(-k * (torch.log(torch.tensor(10000.0)) / (half - 1)))
    arg = t[:, None] * scale[None, :]
    return torch.cat([torch.sin(arg), torch.cos(arg)], dim=-1)


# ------------------------------------------------------------------------------------------
# UNet forward (three variants), functional over a state_dict.
# v1: first-iteration/unet.py:12-115; v2: second-iteration/unet.py:12-158; v3: third-iteration/unet.py:94-178
# ------------------------------------------------------------------------------------------
class _Net:
    def __init__(self, sd, variant, train, bufs):
        self.sd, self.v, self.train, self.bufs = sd, variant, train, bufs

    def conv(self, x, key, pad):
        return F.conv2d(x, self.sd[key + ".weight"], self.sd[key + ".bias"], padding=pad)

    def bn(self, x, key):
        rm, rv = self.bufs[key + ".running_mean"], self.bufs[key + ".running_var"]
        y = F.batch_norm(x, rm, rv, self.sd[key + ".weight"], self.sd[key + ".bias"],
                         training=self.train, momentum=0.1, eps=1e-5)
        if self.train:
            self.bufs[key + ".num_batches_tracked"] = self.bufs[key + ".num_batches_tracked"] + 1
        return y

    def double_conv(self, x, key):
        p = key + ".double_conv"
        if self.v == 1:                                    # (conv-BN-ReLU) x 2
            x = F.relu(self.bn(self.conv(x, p + ".0", 1), p + ".1"))
            return F.relu(self.bn(self.conv(x, p + ".3", 1), p + ".4"))
        x = F.gelu(self.bn(self.conv(x, p + ".0", 1), p + ".1"))        # conv-BN-GELU
        r = p + ".3.conv_block"                                          # residual block
        h = F.gelu(self.bn(self.conv(x, r + ".0", 1), r + ".1"))
        h = self.bn(self.conv(h, r + ".3", 1), r + ".4")
        return F.gelu(x + h)

    def attention(self, x, key):
        B, C, H, W = x.shape
        q = self.conv(x, key + ".query", 0).view(B, -1, H * W).permute(0, 2, 1)     # [B,N,dq]
        k = self.conv(x, key + ".key", 0).view(B, -1, H * W)                        # [B,dq,N]
        a = F.softmax(torch.bmm(q, k), dim=-1)                                      # [B,N,N] no 1/sqrt(d)
        v = self.conv(x, key + ".value", 0).view(B, -1, H * W)                      # [B,C,N]
        o = torch.bmm(v, a.permute(0, 2, 1)).view(B, C, H, W)
        return self.sd[key + ".gamma"] * o + x

    def down(self, x, key):
        x = self.double_conv(x, key + ".conv")
        if self.v >= 2:
            x = self.attention(x, key + ".attention")
        return x, F.max_pool2d(x, 2, 2)

    def up(self, x, skip, key):
        x = F.conv_transpose2d(x, self.sd[key + ".up.weight"], self.sd[key + ".up.bias"], stride=2)
        dy, dx = skip.shape[2] - x.shape[2], skip.shape[3] - x.shape[3]
        x = F.pad(x, [dx // 2, dx - dx // 2, dy // 2, dy - dy // 2])     # negative = crop
        x = self.double_conv(torch.cat([skip, x], dim=1), key + ".conv")
        if self.v >= 2:
            x = self.attention(x, key + ".attention")
        return x


def unet_forward(sd, variant, x, t, y=None, train=False):
    """Returns (eps_hat, feature_maps[7], buffers_after). `sd` maps the reference's state_dict
    keys to CPU tensors; buffers are never modified in place (a train-mode call returns the
    updated running statistics in buffers_after)."""
    bufs = {k: v.clone() for k, v in sd.items() if "running_" in k or k.endswith("num_batches_tracked")}
    n = _Net(sd, variant, train, bufs)
    emb_dim = sd["time_fc1.weight"].shape[1]
    te = time_embedding(t, emb_dim).to(x.dtype)       # (a float64 state_dict and input give a float64 restatement)
    temb = [F.linear(te, sd[f"time_fc{i}.weight"], sd[f"time_fc{i}.bias"])[:, :, None, None] for i in (1, 2, 3, 4)]
    if y is not None:                                                    # third-iteration/unet.py:135-145
        le = F.embedding(y, sd["label_embedding.weight"])
        le = le.view(le.size(0), -1)
        for i in (1, 2, 3, 4):
            temb[i - 1] = temb[i - 1] + F.linear(le, sd[f"label_emdedding_fc{i}.weight"],
                                                 sd[f"label_emdedding_fc{i}.bias"])[:, :, None, None]
    fms = []
    skip1 = n.double_conv(x, "enc1") + temb[0]
    fms.append(skip1)
    skip2, h = n.down(skip1, "enc2")
    skip2 = skip2 + temb[1]
    fms.append(skip2)
    skip3, h = n.down(h, "enc3")                                         # pooled tensor carries no temb
    skip3 = skip3 + temb[2]
    fms.append(skip3)
    if variant == 1:
        h = n.double_conv(h, "bottleneck")
        fms.append(h)                                                    # v1 records before the add
        h = h + temb[3]
    else:
        h = n.attention(n.double_conv(h, "bottleneck.0"), "bottleneck.1")
        h = h + temb[3]
        fms.append(h)
    for key, skip, te_k in (("dec1", skip3, temb[2]), ("dec2", skip2, temb[1]), ("dec3", skip1, temb[0])):
        h = n.up(h, skip, key)
        if variant == 1:
            fms.append(h)
            h = h + te_k
        else:
            h = h + te_k
            fms.append(h)
    out = F.conv2d(h, sd["final_conv.weight"], sd["final_conv.bias"])
    return out, fms, bufs


# ------------------------------------------------------------------------------------------
# Sampling loop -- first-iteration/diffusion.py:17-42, third-iteration/diffusion.py:19-53
# ------------------------------------------------------------------------------------------
def sample(model_fn, betas, x_T, noises, labels=None, cfg_scale=0.0, record=None):
    """Ancestral sampling with injected noise. model_fn(x, t_tensor, labels_or_None) -> eps.
    noises[i] is the draw of the i-th step with t > 0 (t = T-1 first). Returns the reference's
    stage list [(x_t copy, t)] at t % (T//10) == 0 or t == 0."""
    betas = np.asarray(betas, f32)
    T = len(betas)
    alphas, abar = schedule_tables(betas)
    x = torch.as_tensor(x_T).clone()
    stages, zi = [], 0
    for t in reversed(range(T)):
        tt = torch.full((x.shape[0],), t, dtype=torch.long)
        eps = model_fn(x, tt, labels)
        if cfg_scale > 0:
            eps_u = model_fn(x, tt, None)
            eps = torch.from_numpy(cfg_blend(eps_u.numpy(), eps.numpy(), cfg_scale))
        if record is not None:
            record.append(eps.clone())
        if t > 0:
            z = np.asarray(noises[zi], f32)
            zi += 1
        else:
            z = np.zeros(tuple(x.shape), f32)
        c1, c2, c3 = step_coefficients(betas, alphas, abar, t)
        x = torch.from_numpy(reverse_step(x.numpy(), eps.numpy(), z, c1, c2, c3))
        if t % (T // 10) == 0 or t == 0:
            stages.append((x.clone(), t))
    return stages


# ------------------------------------------------------------------------------------------
# Training-loop body -- first-iteration/main.py:65-87 (Adam lr 2e-4, eps-MSE)
# ------------------------------------------------------------------------------------------
PARAM_SUFFIXES = (".weight", ".bias", ".gamma")


def is_param(key):
    return key.endswith(PARAM_SUFFIXES)


def loss_and_grads(sd, variant, x, t, target, y=None):
    """MSE(unet(x,t[,y]) in train mode, target) and d loss / d every parameter (torch autograd on CPU)."""
    work = {k: (v.detach().clone().requires_grad_(True) if is_param(k) else v) for k, v in sd.items()}
    out, fms, bufs = unet_forward(work, variant, x, t, y, train=True)
    loss = F.mse_loss(out, target)
    names = [k for k in work if is_param(k)]
    grads = torch.autograd.grad(loss, [work[k] for k in names], allow_unused=True)
    g = {k: (gr if gr is not None else torch.zeros_like(work[k])) for k, gr in zip(names, grads)}
    return loss.detach(), g, out.detach(), bufs


def adam_step(params, grads, state, lr=2e-4, b1=0.9, b2=0.999, eps=1e-8):
    """torch.optim.Adam defaults (no weight decay, no amsgrad), fp32 state."""
    state["step"] = state.get("step", 0) + 1
    s = state["step"]
    for k, p in params.items():
        g = grads[k]
        m = state.setdefault("m::" + k, torch.zeros_like(p))
        v = state.setdefault("v::" + k, torch.zeros_like(p))
        m.mul_(b1).add_(g, alpha=1 - b1)
        v.mul_(b2).addcmul_(g, g, value=1 - b2)
        bc1, bc2 = 1 - b1 ** s, 1 - b2 ** s
        denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
        p.addcdiv_(m, denom, value=-(lr / bc1))


# ---- scorer: the MNIST classifier and the metrics computed from its logits -------------------------
def simple_cnn_forward(sd, x):
    """SimpleCNN.forward, third-iteration/classifier.py:18-24: logits [B, 10] of [B, 1, 32, 32] images."""
    x = F.max_pool2d(F.relu(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)), 2, 2)
    x = F.max_pool2d(F.relu(F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], padding=1)), 2, 2)
    x = x.view(x.size(0), -1)
    x = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
    return F.linear(x, sd["fc2.weight"], sd["fc2.bias"])


def inception_split_scores(logits, splits=10):
    """Per-split exp(mean KL) terms of the Inception Score, third-iteration/metrics2.py:77,83-87 -> fp32 [splits]."""
    preds = F.softmax(logits, dim=1)
    n = preds.size(0)
    out = []
    for k in range(splits):
        part = preds[k * (n // splits):(k + 1) * (n // splits), :]
        py = torch.mean(part, dim=0)
        kl_div = part * (torch.log(part) - torch.log(py.unsqueeze(0)))
        out.append(torch.exp(torch.mean(torch.sum(kl_div, dim=1))))
    return torch.stack(out)


def inception_score(logits, splits=10):
    """(mean, std) over the splits, third-iteration/metrics2.py:89."""
    s = inception_split_scores(logits, splits)
    return torch.mean(s), torch.std(s)


def fid(act_real, act_gen):
    """Frechet distance between two sets of feature rows (numpy), third-iteration/metrics2.py:107-117."""
    from scipy.linalg import sqrtm
    mu_real, sigma_real = np.mean(act_real, axis=0), np.cov(act_real, rowvar=False)
    mu_gen, sigma_gen = np.mean(act_gen, axis=0), np.cov(act_gen, rowvar=False)
    diff = mu_real - mu_gen
    covmean = sqrtm(sigma_real @ sigma_gen)
    if isinstance(covmean, tuple):
        covmean = covmean[0]
    if np.iscomplexobj(covmean):
        covmean = covmean.real
    return float(np.sum(diff ** 2) + np.trace(sigma_real + sigma_gen - 2 * covmean))


# ---- hand-off to the callers -------------------------------------------------------------------------
def to_images(x):
    """(x + 1) / 2 then clamp to [0, 1], first-iteration/generate_samples.py:36-37 (torch tensor in, tensor out)."""
    return torch.clamp((x + 1) / 2, 0, 1)


def make_grid(tensor, nrow=8, padding=2, normalize=False, pad_value=0.0):
    """Restatement of torchvision.utils.make_grid (torchvision 0.2x, un-pinned in requirements.txt:4; not installed
    in the build image, so this function is pinned by no fixture: parity unpinned) for a 4-D tensor with the default
    value_range / scale_each, as first-iteration/generate_samples.py:71,93 call it."""
    if tensor.dim() == 4 and tensor.size(1) == 1:
        tensor = torch.cat((tensor, tensor, tensor), 1)
    if normalize:
        tensor = tensor.clone()
        low, high = float(tensor.min()), float(tensor.max())
        tensor.clamp_(min=low, max=high)
        tensor.sub_(low).div_(max(high - low, 1e-5))
    if tensor.size(0) == 1:
        return tensor.squeeze(0)
    nmaps = tensor.size(0)
    xmaps = min(nrow, nmaps)
    ymaps = int(math.ceil(float(nmaps) / xmaps))
    height, width = int(tensor.size(2) + padding), int(tensor.size(3) + padding)
    grid = tensor.new_full((tensor.size(1), height * ymaps + padding, width * xmaps + padding), pad_value)
    k = 0
    for y in range(ymaps):
        for x in range(xmaps):
            if k >= nmaps:
                break
            grid.narrow(1, y * height + padding, height - padding).narrow(2, x * width + padding, width - padding).copy_(tensor[k])
            k = k + 1
    return grid
