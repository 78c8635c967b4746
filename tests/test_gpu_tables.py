"""Host-side contracts of the library checked ON THE GPU BOX (the tables come from the box's own
libm and compiler runtime, so bit-exactness has to be shown there too), plus behaviour at the
module boundary that needs a device: schedule tables created on the GPU, labels outside the
embedding table, eval after a train-mode forward without an optimizer step, add_noise without a
host synchronisation."""
import pytest
import torch

import test_host_cabi as H
from conftest import load_golden
from gpu_util import assert_close
from oracle import ddpm_oracle as O
from seeding import seeded_state_dict, seeded_tensor
import diffusion_model_amd as pkg

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("kind", ["linear", "quadratic", "cosine"])
@pytest.mark.parametrize("T", [10, 100, 1000])
def test_tables_bit_exact_on_the_box(kind, T):
    H.test_tables_bit_exact(kind, T)
    # the same tables when the caller asks for them on the GPU (first-iteration/main.py:22-23 passes device)
    g = load_golden("schedules.npz")
    betas = pkg.BetaScheduler(T, device="cuda").get_schedule(kind)
    assert betas.is_cuda and betas.cpu().numpy().tobytes() == g[f"{kind}_{T}_betas"].tobytes()
    dm = pkg.make_diffusion_class(3)(None, betas, T)
    assert dm.alpha_scheduler.is_cuda and dm.alpha_bar_scheduler.is_cuda
    assert dm.alpha_scheduler.cpu().numpy().tobytes() == g[f"{kind}_{T}_alphas"].tobytes()
    assert dm.alpha_bar_scheduler.cpu().numpy().tobytes() == g[f"{kind}_{T}_alpha_bar"].tobytes()


def test_step_coefficients_on_the_box():
    H.test_step_coefficients_match_ieee_and_reference_within_ulp()


@pytest.mark.parametrize("variant", [1, 2, 3])
def test_state_dict_contract_on_the_box(variant):
    H.test_state_dict_contract(variant)
    U = pkg.make_unet_class(variant)
    m = (U(1, 1, 8, 16) if variant < 3 else U(3, 3, 8, 16, num_classes=10)).cuda()
    keys = [k for k, _ in m.state_dict().items()]
    assert all(v.is_cuda for v in m.state_dict().values()) and len(keys) == len(set(keys))


def test_no_cpu_fallback_on_the_box():
    H.test_no_cpu_fallback()
    H.test_library_exports_every_declared_symbol()


def test_named_schedules_and_bad_name_on_the_box():
    H.test_named_schedule_methods_and_bad_name()


def _tiny_v3(seed=4):
    m = pkg.make_unet_class(3)(1, 1, 8, 16, num_classes=10)
    sd = seeded_state_dict(m.state_dict(), seed)
    m.load_state_dict(sd)
    return m.cuda(), sd


@pytest.mark.parametrize("bad", [10, -1, 1 << 40])
def test_label_outside_the_embedding_table_raises(bad):
    """nn.Embedding raises for such a label (third-iteration/unet.py:136); nothing may be read or written out of bounds."""
    m, _ = _tiny_v3()
    x, t = seeded_tensor((2, 1, 28, 28), 1, "x").cuda(), torch.tensor([5, 900]).cuda()
    m.eval()
    with torch.no_grad():
        with pytest.raises(IndexError, match="label out of range"):
            m(x, t, torch.tensor([3, bad]).cuda())
        with pytest.raises(IndexError, match="label out of range"):
            m(x, t, torch.tensor([bad]).cuda())
        ok = m(x, t, torch.tensor([3, 9]).cuda())          # the largest valid label still works afterwards
    assert torch.isfinite(ok).all()
    # a training step does not wait for the device: the bad sample's prediction is NaN (the kernels never dereference
    # the label; train-mode BatchNorm may spread the NaN over the batch) and the error is raised by the next call that
    # finds the verdict arrived
    import warnings
    m.train()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore", RuntimeWarning)      # (the NaN also takes the network off the fp16 pair)
        out = m(x, t, torch.tensor([bad, 0]).cuda())
        assert torch.isnan(out[0]).all()
        torch.cuda.synchronize()
        with pytest.raises(IndexError, match="label out of range"):
            m(x, t, torch.tensor([1, 0]).cuda())
        assert torch.isfinite(m(x, t, torch.tensor([1, 0]).cuda())).all()     # reported once


def test_training_step_with_fresh_labels_does_not_synchronise():
    """A conditional training step (third-iteration/main.py:91 with labels) gets a new label tensor every iteration;
    neither the forward nor the backward may make the host wait for the GPU."""
    m, _ = _tiny_v3()
    m.train()
    x, t = seeded_tensor((4, 1, 28, 28), 1, "x").cuda(), torch.tensor([5, 900, 17, 0]).cuda()
    tgt = seeded_tensor((4, 1, 28, 28), 2, "tgt").cuda()
    labels = [torch.tensor(v).cuda() for v in ([1, 2, 3, 4], [9, 0, 5, 5], [7, 7, 1, 0])]
    crit = torch.nn.MSELoss()
    crit(m(x, t, labels[0]), tgt).backward()                # first step: caches, workspaces, pinned buffers
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        for y in labels[1:]:
            m.zero_grad()
            crit(m(x, t, y), tgt).backward()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters())


def test_label_kernels_never_dereference_a_bad_label():
    """Straight through the C ABI (no host check in the way): the forward marks the sample's rows NaN and the
    backward adds nothing to the table; valid samples of the same batch are untouched."""
    import ctypes as C
    from diffusion_model_amd._lib import EmbedArgs, EmbedBwdArgs
    from gpu_util import call, rnd, stream
    B, E, chans, ncls = 3, 16, [8, 16, 32, 64], 4
    t = torch.tensor([1, 2, 3]).cuda()
    table = rnd((ncls, E), 1).cuda()

    def run(labels):
        y = torch.tensor(labels).cuda()
        a = EmbedArgs()
        a.batch, a.emb_dim, a.n_labels, a.t_is_float, a.n_classes = B, E, B, 0, ncls
        a.t, a.y, a.label_table = t.data_ptr(), y.data_ptr(), table.data_ptr()
        ws = [rnd((c, E), 10 + k, 0.3).cuda() for k, c in enumerate(chans)]
        bs = [rnd((c,), 20 + k).cuda() for k, c in enumerate(chans)]
        outs = [torch.empty((B, c), device="cuda") for c in chans]
        emb = torch.empty((B, E), device="cuda")
        for k in range(4):
            a.ch[k], a.w[k], a.b[k], a.lw[k], a.lb[k], a.out[k] = chans[k], ws[k].data_ptr(), bs[k].data_ptr(), ws[k].data_ptr(), \
                bs[k].data_ptr(), outs[k].data_ptr()
        a.emb_out = emb.data_ptr()
        call("ddpm_time_embed", C.byref(a), stream())
        g = EmbedBwdArgs()
        g.batch, g.emb_dim, g.n_labels, g.n_classes = B, E, B, ncls
        dts = [torch.ones((B, c), device="cuda") for c in chans]
        dw = [torch.empty_like(w) for w in ws] + [torch.empty_like(w) for w in ws]
        db = [torch.empty_like(b) for b in bs] + [torch.empty_like(b) for b in bs]
        dtab = torch.zeros((ncls, E), device="cuda")
        g.emb, g.y, g.label_table, g.dlabel_table = emb.data_ptr(), y.data_ptr(), table.data_ptr(), dtab.data_ptr()
        for k in range(4):
            g.ch[k], g.dtemb[k], g.lw[k] = chans[k], dts[k].data_ptr(), ws[k].data_ptr()
            g.dw[k], g.db[k], g.dlw[k], g.dlb[k] = dw[k].data_ptr(), db[k].data_ptr(), dw[4 + k].data_ptr(), db[4 + k].data_ptr()
        call("ddpm_time_embed_bwd", C.byref(g), stream())
        torch.cuda.synchronize()
        return [o.cpu() for o in outs], dtab.cpu()

    good, tab_good = run([0, 3, 2])
    bad, tab_bad = run([0, 4, 2])              # label 4 == n_classes: one past the table
    for og, ob in zip(good, bad):
        assert torch.equal(og[0], ob[0]) and torch.equal(og[2], ob[2])
        assert torch.isnan(ob[1]).all() and torch.isfinite(og[1]).all()
    assert torch.equal(tab_bad[0], tab_good[0]) and torch.equal(tab_bad[2], tab_good[2])
    assert torch.all(tab_bad[3] == 0) and torch.any(tab_good[3] != 0)


@pytest.mark.parametrize("with_label", [False, True])
@pytest.mark.parametrize("batch", [1, 5, 512])
def test_embedding_of_a_uniform_step_equals_the_per_sample_one(batch, with_label):
    """A sampling step hands every sample the same t (and at most one label, third-iteration/diffusion.py:34-37): the
    embedding rows are then computed once and replicated, and must be the bits the per-sample kernel gives."""
    m, _ = _tiny_v3()
    eng = m._engine
    t = torch.full((batch,), 737, dtype=torch.long, device="cuda")
    y = torch.tensor([7]).cuda() if with_label else None
    emb_a, emb_b = (torch.full((batch, eng.emb), float("nan"), device="cuda") for _ in range(2))
    per_sample = eng.time_embed(t, y, emb_out=emb_a, uniform=False)
    once = eng.time_embed(t, y, emb_out=emb_b, uniform=True)
    assert torch.equal(emb_a, emb_b) and torch.isfinite(emb_a).all()
    for a, b in zip(per_sample, once):
        assert a.shape == b.shape and torch.equal(a, b) and torch.isfinite(a).all()
    # per-sample labels are not uniform whatever the flag says: the mirror falls back to the per-sample kernel
    if with_label and batch > 1:
        ys = (torch.arange(batch) % 10).cuda()
        a = eng.time_embed(t, ys, uniform=False)
        b = eng.time_embed(t, ys, uniform=True)
        assert all(torch.equal(u, v) for u, v in zip(a, b)) and not torch.equal(a[0][0], a[0][1])


def test_eval_after_a_train_forward_without_optimizer_step_uses_fresh_running_statistics():
    """eval -> train-mode forward under no_grad (updates running_mean / running_var in place) -> eval: the second eval
    must see the updated statistics (the folded scale/shift pairs are cached)."""
    m = pkg.make_unet_class(2)(1, 1, 8, 16)
    sd = seeded_state_dict(m.state_dict(), 6)
    m.load_state_dict(sd)
    m = m.cuda()
    x, t = seeded_tensor((4, 1, 28, 28), 2, "x"), torch.tensor([0, 10, 500, 999])
    with torch.no_grad():
        m.eval()
        first = m(x.cuda(), t.cuda()).cpu()
        m.train()
        m(x.cuda() * 3.0 + 1.0, t.cuda())                  # no optimizer step follows
        m.eval()
        second = m(x.cuda(), t.cuda()).cpu()
    want_first = O.unet_forward(sd, 2, x, t, None, train=False)[0]
    _, _, bufs = O.unet_forward(sd, 2, x * 3.0 + 1.0, t, None, train=True)
    sd2 = dict(sd)
    sd2.update(bufs)
    want_second = O.unet_forward(sd2, 2, x, t, None, train=False)[0]
    assert_close(first, want_first, 5e-5, "eval before")
    assert_close(second, want_second, 5e-5, "eval after the statistics moved")
    assert (want_second - want_first).abs().max() > 1e-3   # the case is not vacuous: the statistics did move the output


def test_add_noise_with_device_timesteps_does_not_synchronise():
    betas = pkg.BetaScheduler(1000, device="cuda").get_schedule("linear")
    dm = pkg.make_diffusion_class(2)(None, betas, 1000)
    x, n = torch.randn(8, 1, 28, 28, device="cuda"), torch.randn(8, 1, 28, 28, device="cuda")
    t = torch.randint(0, 1000, (8,), device="cuda")
    dm.add_noise(x, n, t)                                   # first call places the table on the device
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("error")
    try:
        out = dm.add_noise(x, n, t)
    finally:
        torch.cuda.set_sync_debug_mode("default")
    abar = dm.alpha_bar_scheduler[t].view(-1, 1, 1, 1)
    assert_close(out.cpu(), (abar.sqrt() * x + (1 - abar).sqrt() * n).cpu(), 1e-6, "add_noise")
    with pytest.raises(IndexError):                         # host timesteps are still range-checked (free there)
        dm.add_noise(x, n, torch.full((8,), 1000))
