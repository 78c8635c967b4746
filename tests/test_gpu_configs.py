"""Shapes of the remaining benchmark configurations as parity cases, against the CPU oracle on
seeded inputs: colour 3x32x32 training at full width (data-parallel training config) and 3x64x64
with attention at N = 4096 / 1024 / 256 (roofline-stress config); plus a size-independent
property at full width: an eval-mode forward does not depend on how the batch is cut, bit for bit
(every output element is a fixed-order sum over its own sample's data, whatever tile it lands in).

Tolerances as in the golden-vector tests: forward 5e-5 of tensor scale (1e-4 in train mode),
gradients 2e-3 of each tensor's max + 5e-6.
"""
import pytest
import torch

from gpu_util import assert_close
from oracle import ddpm_oracle as O
from seeding import seeded_ints, seeded_state_dict, seeded_tensor
import diffusion_model_amd as pkg

pytestmark = pytest.mark.gpu


def build(variant, cin, base, emb, seed, classes=None):
    U = pkg.make_unet_class(variant)
    m = U(cin, cin, base, emb, num_classes=classes) if variant == 3 and classes else U(cin, cin, base, emb)
    m.load_state_dict(seeded_state_dict(m.state_dict(), seed))
    return m.cuda()


def cpu_sd(m):
    return {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}


def test_colour_64x64_eval_and_gradients():
    """3x64x64, v3 with labels: attention at N = 4096 (fp32 kernel, 16 channels), 1024, and 256 (split-bf16 kernel)."""
    m = build(3, 3, 16, 32, 11, classes=10)
    sd = cpu_sd(m)
    x = seeded_tensor((2, 3, 64, 64), 1, "x")
    t = torch.tensor([17, 903])
    y = torch.tensor([3, 8])
    m.eval()
    with torch.no_grad():
        out = m(x.cuda(), t.cuda(), y.cuda())
    want, fms, _ = O.unet_forward(sd, 3, x, t, y, train=False)
    assert_close(out.cpu(), want, 5e-5, "64x64 eval eps")
    for i, fm in enumerate(m.feature_maps):
        assert_close(fm.cpu(), fms[i], 5e-5, f"64x64 eval fm{i}")
    m.train()
    tgt = seeded_tensor((2, 3, 64, 64), 2, "tgt")
    loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda(), y.cuda()), tgt.cuda())
    loss.backward()
    ref_loss, grads, _, _ = O.loss_and_grads(sd, 3, x, t, tgt, y)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:6]


def test_colour_64x64_full_width_eval():
    """The roofline-stress configuration at its real width: (3,3,64,128,num_classes=10) on 3x64x64. Attention runs at
    N = 4096 with 64 and 128 channels and at N = 1024 with 256 (the split-bf16 kernels), and at N = 256 with 512
    channels; eps-hat and all seven feature maps against the CPU oracle."""
    m = build(3, 3, 64, 128, 13, classes=10).eval()
    sd = cpu_sd(m)
    x = seeded_tensor((2, 3, 64, 64), 3, "x")
    t = torch.tensor([801, 12])
    y = torch.tensor([7, 0])
    with torch.no_grad():
        out = m(x.cuda(), t.cuda(), y.cuda())
        want, fms, _ = O.unet_forward(sd, 3, x, t, y, train=False)
    assert_close(out.cpu(), want, 5e-5, "64x64 full-width eval eps")
    for i, fm in enumerate(m.feature_maps):
        assert_close(fm.cpu(), fms[i], 5e-5, f"64x64 full-width eval fm{i}")


@pytest.mark.parametrize("hw", [32, 28])
def test_base_channels_128_eval_and_gradients(hw):
    """The wide colour model the reference's CIFAR scripts build (UNet(3, 3, 128, 128), second-iteration/main.py:45):
    1024 channels and 128 query channels in the bottleneck, a fused q|k|v projection of 1280 columns."""
    m = build(2, 3, 128, 128, 14)
    sd = cpu_sd(m)
    x, tgt = seeded_tensor((2, 3, hw, hw), 41, "x"), seeded_tensor((2, 3, hw, hw), 141, "tgt")
    t = torch.tensor([650, 3])
    m.eval()
    with torch.no_grad():
        out = m(x.cuda(), t.cuda())
        want, fms, _ = O.unet_forward(sd, 2, x, t, None, train=False)
    assert_close(out.cpu(), want, 5e-5, "base-128 eval eps")
    for i, fm in enumerate(m.feature_maps):
        assert_close(fm.cpu(), fms[i], 5e-5, f"base-128 eval fm{i}")
    m.train()
    loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda()), tgt.cuda())
    loss.backward()
    ref_loss, grads, _, _ = O.loss_and_grads(sd, 2, x, t, tgt)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:6]
    with pytest.raises(NotImplementedError, match="above 128"):
        pkg.make_unet_class(2)(3, 3, 160, 128)


def test_colour_32x32_full_width_train_step():
    """(3,3,64,128) on 3x32x32: loss and every parameter gradient of one eps-MSE step.

    The input seed is one for which the CPU oracle itself is stable: with random full-width weights the
    2x2 max-pool windows hold near-ties, and a tie that flips under a 1e-6 relative perturbation of the
    weights re-routes a gradient and moves the encoder gradients by up to 5 % (seen with other seeds in
    the oracle alone). Under such a perturbation this seed stays within a quarter of the tolerance.
    """
    m = build(2, 3, 64, 128, 12).train()
    sd = cpu_sd(m)
    x, tgt = seeded_tensor((4, 3, 32, 32), 34, "x"), seeded_tensor((4, 3, 32, 32), 134, "tgt")
    t = seeded_ints((4,), 1000, 234, "t")
    loss = torch.nn.MSELoss()(m(x.cuda(), t.cuda()), tgt.cuda())
    loss.backward()
    ref_loss, grads, _, _ = O.loss_and_grads(sd, 2, x, t, tgt)
    assert abs(loss.item() - ref_loss.item()) <= 1e-4 * ref_loss.item()
    worst = []
    for k, p in m.named_parameters():
        ref = grads[k]
        err = (p.grad.cpu() - ref).abs().max().item()
        worst.append((err / (2e-3 * ref.abs().max().item() + 5e-6), k, err))
    worst.sort(reverse=True)
    assert worst[0][0] <= 1.0, worst[:6]


@pytest.mark.parametrize("variant,hw", [(1, 28), (2, 28), (3, 32)])
def test_eval_forward_is_independent_of_the_batch_cut(variant, hw):
    m = build(variant, 1, 64, 128, 20 + variant, classes=10 if variant == 3 else None).eval()
    x = seeded_tensor((24, 1, hw, hw), 6, "x").cuda()
    t = seeded_ints((24,), 1000, 7, "t").cuda()
    y = seeded_ints((24,), 10, 8, "y").cuda() if variant == 3 else None
    with torch.no_grad():
        def run(sl):
            return m(x[sl], t[sl], y[sl]) if variant == 3 else m(x[sl], t[sl])
        whole = run(slice(0, 24))
        parts = torch.cat([run(slice(0, 5)), run(slice(5, 16)), run(slice(16, 24))])
    assert torch.equal(whole, parts)


def test_two_stream_backward_matches_one_stream():
    """The weight gradients run on a second stream beside the data-gradient chain: same gradients as the
    single-stream order (weight gradients are split-K atomic sums, so equal to rounding, not bit for bit),
    step after step (the second step reuses memory the first one released)."""
    from diffusion_model_amd.engine_train import TrainMixin
    x = seeded_tensor((8, 1, 28, 28), 5, "x")
    eps = seeded_tensor((8, 1, 28, 28), 6, "eps")
    t = torch.tensor([3, 999, 500, 17, 0, 250, 731, 64])
    results = []
    saved = TrainMixin._use_side
    try:
        for side in (False, True):
            TrainMixin._use_side = side
            m = build(2, 1, 32, 64, 21)
            m.train()
            per_step = []
            for _ in range(3):
                m.zero_grad()
                loss = torch.nn.functional.mse_loss(m(x.cuda(), t.cuda()), eps.cuda())
                loss.backward()
                torch.cuda.synchronize()
                per_step.append({k: p.grad.detach().cpu().clone() for k, p in m.named_parameters()})
            results.append(per_step)
    finally:
        TrainMixin._use_side = saved
    for a, b in zip(*results):
        for k in a:
            assert_close(b[k], a[k], 2e-5, f"two-stream gradient {k}")
