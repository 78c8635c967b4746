"""Schedule of HIP kernels that evaluates the three UNet variants (and their backward pass).

The nn.Module in unet_impl.py only holds parameters under the reference's state_dict names; all
arithmetic happens here, as an explicit list of calls into libddpm_mi355x.so on raw device
pointers. Activations are NHWC fp32 (torch channels_last memory), allocated through torch's
caching allocator; kernels run on torch's current stream.

Reference behaviour this schedule reproduces: first-iteration/unet.py:75-115,
second-iteration/unet.py:118-158, third-iteration/unet.py:126-178.
"""
import ctypes as C
import os
import warnings
import weakref

import torch

from . import _lib
from ._lib import ConvDesc, EmbedArgs, Epilogue, call, conv_math, math_scope, packed_floats
from .engine_train import TrainMixin, UNetFunction

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
BN_EPS = 1e-5
# DDPM_ATTN_FUSED=0 (read once): inference-time attention blocks run as ddpm_conv1x1_fwd + ddpm_attention_fwd, as training
# calls do, instead of ddpm_self_attention_infer: a diagnostic switch for A/B measurements (the results are the same bits).
_FUSED_ATTENTION = os.environ.get("DDPM_ATTN_FUSED", "1") != "0"


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack_1x1(w, transpose):
    """[rows][cin] fp32 weight of a 1x1 conv (or its transpose: the data-gradient GEMM) -> the library's packed layout."""
    rows, cin = (w.shape[1], w.shape[0]) if transpose else (w.shape[0], w.shape[1])
    out = torch.empty(packed_floats(1, rows, cin), device=w.device, dtype=torch.float32)
    call("ddpm_pack_conv1x1", _ptr(w.contiguous()), rows, cin, int(transpose), _ptr(out), _stream())
    return out


def _round_up(a, b):
    return (a + b - 1) // b * b


class View:
    """A channel slice [c0, c0+c) of an NHWC buffer `buf` of shape [B, H, W, ld]."""

    __slots__ = ("buf", "c0", "c")

    def __init__(self, buf, c0=0, c=None):
        self.buf, self.c0 = buf, c0
        self.c = buf.shape[-1] - c0 if c is None else c

    @property
    def ld(self):
        return self.buf.shape[-1]

    @property
    def ptr(self):
        return C.c_void_p(self.buf.data_ptr() + 4 * self.c0)

    @property
    def bhw(self):
        return self.buf.shape[0], self.buf.shape[1], self.buf.shape[2]

    def nchw(self):
        """Logical [B, C, H, W] tensor sharing this memory (channels_last strides)."""
        return self.buf[..., self.c0:self.c0 + self.c].permute(0, 3, 1, 2)


class UNetEngine(TrainMixin):
    def __init__(self, module, variant, in_channels, out_channels, base_channels, embedding_dim, num_classes):
        if base_channels % 4 != 0:
            raise NotImplementedError("base_channels must be a multiple of 4 for the 16-byte vector paths")
        if base_channels > 128:
            raise NotImplementedError("base_channels above 128 is not supported (the BatchNorm reductions take at most 1024 channels "
                                      "and the attention blocks at most 128 query channels)")
        if embedding_dim % 2 != 0 or embedding_dim < 4:
            raise NotImplementedError("embedding_dim must be even and >= 4")
        self.m = module
        self.v = variant
        self.cin, self.cout_final = in_channels, out_channels
        self.base, self.emb = base_channels, embedding_dim
        self.num_classes = num_classes
        self._cache = {}
        self._tensors = None
        # Arithmetic of this network's MFMA-fed kernels: None = the library's current mode (DDPM_CONV_MATH, default the
        # fp16 pair); 'bf16x3' once an operand beyond fp16's range has been seen (range_fallback), or set by the caller.
        self.math = None
        self.range_fallbacks = 0       # how often this network has left the fp16 pair because of such an operand
        self._range_pending = None     # (event, slot): a flag read-back enqueued and not yet examined
        self._range_unread = False     # watched kernels of this network were enqueued after its last read-back
        self._range_external = False   # a caller (sample()) reads the flag itself: run() neither clears nor posts
        self._range_host = None        # pinned int32 words the read-backs land in
        self._labels_pending = []      # deferred label range checks of training steps: (event, pinned verdict, classes)
        self.prof = None       # when a list: (kernel label, flops, start event, end event) per sampled MFMA launch
        self.prof_stride = 1   # every prof_stride-th MFMA launch is bracketed by events (a stride that does not
        self._prof_i = 0       # divide the launches per step walks through all launch sites over the steps)

    def _gemm_label(self, kind, m, n):
        """Name of the kernel family the library uses for a forward/data-gradient contraction (profiling label)."""
        math = conv_math()
        if math == "fp32":
            return f"conv_gemm_fp32<{kind}>"
        tag = "split-bf16" if math == "bf16x3" else "fp16-pair"
        if kind == "conv3x3" and n > 32:
            return f"conv3x3_halo<{tag}>"
        return f"conv_gemm_x3<{kind},{tag}>"

    def _launch(self, label, flops, name, *args):
        if self.prof is None:
            return call(name, *args)
        self._prof_i += 1
        if self._prof_i % self.prof_stride:
            return call(name, *args)
        # a bracketed launch runs alone: the other stream of the training backward is drained first and held
        # until the kernel is done, so that the events time this kernel and not its share of a busy chip
        cur, other = torch.cuda.current_stream(), None
        s2 = getattr(self, "_s2", None)
        if s2 is not None:
            other = self._main_stream if cur == s2 else s2
            if other is not None:
                cur.wait_stream(other)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        call(name, *args)
        e1.record()
        if other is not None:
            other.wait_event(e1)
        self.prof.append((label, flops, e0, e1))

    # ------------------------------------------------------------------ entry point
    def invalidate(self):
        """Parameters were moved or replaced (module._apply): drop every derived cache."""
        self._cache.clear()
        self._tensors = None

    # ------------------------------------------------------------------ arithmetic and its range watch
    def math_scope(self):
        """Context in which the library computes in this network's arithmetic."""
        return math_scope(self.math)

    def effective_math(self):
        return self.math or conv_math()

    def range_watching(self):
        """True while this network computes with fp16 pairs, whose operands are clipped beyond +-65504 and whose matrix
        instructions do not hand NaNs on: the kernels raise a device flag when they meet either and the host mirror reacts (DDPM_RANGE_WATCH=0 switches the reaction
        off: results are then clipped silently, as a diagnostic). Not while a HIP graph is being captured: the flag's
        read-back and the event query are not capturable, so a replayed graph runs without the watch."""
        return (self.effective_math() == "f16x2" and os.environ.get("DDPM_RANGE_WATCH", "1") != "0"
                and not torch.cuda.is_current_stream_capturing())

    def _range_slot(self):
        if self._range_host is None:
            self._range_host = torch.zeros(2, dtype=torch.int32).pin_memory()
        return self._range_host

    def range_clear(self):
        """Forget whatever raised the flag before this point of the current stream."""
        call("ddpm_range_flag_read", None, 1, _stream())

    def range_check_sync(self):
        """Read and clear the flag behind everything enqueued so far and WAIT for it: True when one of the kernels
        since the last clear met an operand beyond fp16's range."""
        host = self._range_slot()
        call("ddpm_range_flag_read", C.c_void_p(host.data_ptr()), 1, _stream())
        torch.cuda.current_stream().synchronize()
        return bool(host[0].item())

    def range_post(self):
        """Enqueue a read-and-clear of the flag without waiting; range_poll examines it once the GPU has got there.
        (While one is pending no second one is enqueued: the flag stays raised on the device until it is read.)"""
        if self._range_pending is not None:
            return                     # (_range_unread stays set: a later post picks the flag up)
        host = self._range_slot()
        call("ddpm_range_flag_read", C.c_void_p(host.data_ptr() + 4), 1, _stream())
        ev = torch.cuda.Event()
        ev.record()
        self._range_pending = ev
        self._range_unread = False

    def range_poll(self):
        """Examine a pending read-back if the GPU has finished it (never waits). An operand beyond fp16's range
        moves this network to the three-plane bf16 arithmetic for every later call, with a warning."""
        ev = self._range_pending
        if ev is None or not ev.query():
            return
        self._range_pending = None
        if int(self._range_host[1].item()):
            self.range_fallback("an earlier call")

    def range_fallback(self, where):
        self.math = "bf16x3"
        self.range_fallbacks += 1
        warnings.warn(f"diffusion_model_amd: {where} met activations, weights or gradients beyond +-65504 (or infinite, or NaN), "
                      "which the default fp16-pair arithmetic clips; this network now computes with three bf16 planes per operand "
                      "(full fp32 range and NaN propagation, about 1.5x slower). Set DDPM_CONV_MATH=bf16x3 to start there.",
                      RuntimeWarning, stacklevel=3)

    def _poll_labels(self):
        """Deferred label checks of earlier training steps whose verdict has arrived (never waits)."""
        pending, self._labels_pending = self._labels_pending, []
        for i, (ev, host, n_classes) in enumerate(pending):
            if not ev.query():
                self._labels_pending = pending[i:]
                return
            if int(host.item()):
                self._labels_pending = pending[i + 1:]
                raise IndexError(f"label out of range: an earlier training step was handed a label outside [0, {n_classes}) "
                                 "(the embedding's classes); that sample's outputs and the step's loss were NaN")

    def run(self, x, t, y, training):
        if self._labels_pending and not torch.cuda.is_current_stream_capturing():
            self._poll_labels()
        own_watch = self.range_watching() and not self._range_external
        if own_watch:
            self.range_poll()
            if self._range_pending is None and not self._range_unread:
                self.range_clear()     # the flag is one word per GPU: whatever other work raised it is not this call's
            self._range_unread = True  # watched kernels of this network follow; the next read-back covers them
        with self.math_scope():
            out = self._run(x, t, y, training)
        if own_watch and self.range_watching() and not (torch.is_grad_enabled() and training):
            self.range_post()      # (a differentiated training forward posts after its backward)
        return out

    def _run(self, x, t, y, training):
        if x.dtype != torch.float32:
            x = x.float()
        t = t.to(x.device)
        if y is not None:
            if self.num_classes is None:
                raise AttributeError("this UNet was built without num_classes; it has no label embedding")
            y = y.to(device=x.device, dtype=torch.int64).contiguous()
        named = list(self.m.named_parameters())
        needs_grad = torch.is_grad_enabled() and any(p.requires_grad for _, p in named)
        if x.requires_grad and torch.is_grad_enabled():
            raise NotImplementedError("gradients with respect to the UNet input are not produced (the training loop never asks)")
        if needs_grad:
            if not training:
                raise NotImplementedError("differentiating through eval-mode BatchNorm is not supported; call model.train()")
            names = tuple(n for n, _ in named)
            out, *fms = UNetFunction.apply(self, x, t, y, names, *[p for _, p in named])
            return out, list(fms)
        if training:
            out, fms, _ = self.forward_train(x, t, y)
            return out, fms
        return self.forward_eval(x, t, y)

    def denoise_step(self, x, t, y, step):
        """x_{t-1} from x_t: eval-mode UNet forward whose final 1x1 conv is fused with the reverse step (no gradients)."""
        if x.dtype != torch.float32:
            x = x.float()
        t = t.to(x.device)
        if y is not None:
            if self.num_classes is None:
                raise AttributeError("this UNet was built without num_classes; it has no label embedding")
            y = y.to(device=x.device, dtype=torch.int64).contiguous()
        with torch.no_grad(), self.math_scope():
            return self.forward_eval(x, t, y, want_fms=bool(step.get("want_fms", True)), step=step)

    # ------------------------------------------------------------------ parameters and caches
    def T(self, key):
        if self._tensors is None:
            d = dict(self.m.named_parameters())
            d.update(dict(self.m.named_buffers()))
            self._tensors = d
        return self._tensors[key]

    def _cached(self, kind, keys, build):
        """Value derived from parameters, rebuilt when one of them changes. Packed weights belong to the arithmetic
        they were packed in, so the library's current mode is part of the key (both copies are kept)."""
        ts = [self.T(k) for k in keys]
        stamp = tuple((t._version, t.data_ptr()) for t in ts)
        slot = (kind, keys[0], conv_math())
        hit = self._cache.get(slot)
        if hit is not None and hit[0] == stamp:
            return hit[1]
        with torch.no_grad():
            val = build(*ts)
        self._cache[slot] = (stamp, val)
        return val

    def packed_conv3(self, key, flip=False):
        def build(w):
            rows, depth = (w.shape[1], w.shape[0]) if flip else (w.shape[0], w.shape[1])
            out = torch.empty(packed_floats(9, rows, depth), device=w.device, dtype=torch.float32)
            call("ddpm_pack_conv3x3", _ptr(w.detach().contiguous()), w.shape[0], w.shape[1], int(flip), _ptr(out), _stream())
            return out
        return self._cached("pk3f" if flip else "pk3", (key + ".weight",), build)

    def prepack_conv3(self):
        """Repack every 3x3 weight whose packed copy is stale, all of them in one batched launch. A training step changes
        every weight, and the forward and the backward would otherwise repack them one by one as they reach them (forty
        launches of a few microseconds each). The packed buffers are refilled in place (same stream: kernels enqueued
        earlier still see the old contents)."""
        items = []
        mode = conv_math()
        for (kind, key, packed_in), (stamp, val) in self._cache.items():
            if kind in ("pk3", "pk3f") and packed_in == mode:
                w = self.T(key)
                now = ((w._version, w.data_ptr()),)
                if now != stamp and w.is_contiguous():
                    items.append((kind, key, w, val, now))
        n = len(items)
        if n < 2:
            return
        ptrs = (C.c_void_p * n)(*[it[2].data_ptr() for it in items])
        outs = (C.c_void_p * n)(*[it[3].data_ptr() for it in items])
        cout = (C.c_int * n)(*[it[2].shape[0] for it in items])
        cin = (C.c_int * n)(*[it[2].shape[1] for it in items])
        flip = (C.c_int * n)(*[int(it[0] == "pk3f") for it in items])
        call("ddpm_pack_conv3x3_batch", n, ptrs, cout, cin, flip, outs, _stream())
        for kind, key, _, val, now in items:
            self._cache[(kind, key, mode)] = (now, val)

    def packed_convT(self, key):
        def build(w):
            out = torch.empty(packed_floats(1, 4 * w.shape[1], w.shape[0]), device=w.device, dtype=torch.float32)
            call("ddpm_pack_convT2x2", _ptr(w.detach().contiguous()), w.shape[0], w.shape[1], _ptr(out), _stream())
            return out
        return self._cached("pkT", (key + ".weight",), build)

    def folded_bn(self, conv_key, bn_key):
        """Eval-mode BatchNorm folded with the conv bias: y = conv*scale + shift."""
        def build(cb, w, b, rm, rv):
            scale = w / torch.sqrt(rv + BN_EPS)
            shift = (cb - rm) * scale + b
            return scale.contiguous(), shift.contiguous()
        keys = (conv_key + ".bias", bn_key + ".weight", bn_key + ".bias", bn_key + ".running_mean", bn_key + ".running_var")
        return self._cached("fold", keys, build)

    def packed_qkv(self, key):
        """Fused q|k|v projection: packed [2*dqp + C][C] weight, [2*dqp + C] bias, dqp, and the packed transpose
        (for the data gradient); q and k zero-padded to dqp = round_up(C/8, 8) rows."""
        def build(qw, qb, kw, kb, vw, vb):
            c = vw.shape[0]
            dq = qw.shape[0]
            dqp = _round_up(dq, 8)
            lq = 2 * dqp + c
            dev = vw.device
            wf = torch.empty(packed_floats(1, lq, c), device=dev, dtype=torch.float32)
            wt = torch.empty(packed_floats(1, c, lq), device=dev, dtype=torch.float32)
            b = torch.empty(lq, device=dev, dtype=torch.float32)
            call("ddpm_pack_qkv", _ptr(qw.detach().contiguous()), _ptr(kw.detach().contiguous()), _ptr(vw.detach().contiguous()),
                 _ptr(qb.detach()), _ptr(kb.detach()), _ptr(vb.detach()), c, dq, dqp, _ptr(wf), _ptr(wt), _ptr(b), _stream())
            return wf, b, dqp, wt
        keys = tuple(f"{key}.{n}.{s}" for n in ("query", "key", "value") for s in ("weight", "bias"))
        return self._cached("qkv", keys, build)

    # ------------------------------------------------------------------ kernel wrappers
    @staticmethod
    def _new(b, h, w, c, like):
        return torch.empty((b, h, w, c), device=like.device, dtype=torch.float32)

    def conv3x3(self, src, wkey_packed, cout, out, scale=None, shift=None, residual=None, temb=None, act=ACT_NONE):
        b, h, w = src.bhw
        d = ConvDesc(batch=b, in_h=h, in_w=w, c1=src.c, ld1=src.ld, c2=0, ld2=0, cout=cout, ldo=out.ld, act=act,
                     ldr=residual.ld if residual is not None else 0, out_h=h, out_w=w, off_y=0, off_x=0)
        ep = Epilogue(_ptr(scale), _ptr(shift), residual.ptr if residual is not None else None, _ptr(temb))
        m = b * h * w
        self._launch(self._gemm_label("conv3x3", m, cout), 2.0 * m * cout * 9 * src.c, "ddpm_conv3x3_fwd", C.byref(d), src.ptr,
                     None, _ptr(wkey_packed), C.byref(ep), out.ptr, _stream())

    def conv3x3_stats(self, src, wkey_packed, cout, out, bias, partial):
        """conv3x3 (+bias) into `out` that also leaves the partial sums of train-mode BatchNorm statistics over `out` in
        `partial` (the engine's float64 reduction workspace); returns the number of partial blocks."""
        b, h, w = src.bhw
        d = ConvDesc(batch=b, in_h=h, in_w=w, c1=src.c, ld1=src.ld, c2=0, ld2=0, cout=cout, ldo=out.ld, act=ACT_NONE, ldr=0, out_h=h, out_w=w,
                     off_y=0, off_x=0)
        ep = Epilogue(None, _ptr(bias), None, None)
        m = b * h * w
        nblk = C.c_int(0)
        self._launch(self._gemm_label("conv3x3", m, cout), 2.0 * m * cout * 9 * src.c, "ddpm_conv3x3_fwd_stats", C.byref(d), src.ptr,
                     None, _ptr(wkey_packed), C.byref(ep), out.ptr, _ptr(partial), C.byref(nblk), _stream())
        return nblk.value

    def conv1x1(self, src, w, bias, cout, out):
        b, h, wd = src.bhw
        d = ConvDesc(batch=b, in_h=h, in_w=wd, c1=src.c, ld1=src.ld, c2=0, ld2=0, cout=cout, ldo=out.ld, act=ACT_NONE,
                     ldr=0, out_h=h, out_w=wd, off_y=0, off_x=0)
        ep = Epilogue(None, _ptr(bias), None, None)
        m = b * h * wd
        self._launch(self._gemm_label("conv1x1", m, cout), 2.0 * m * cout * src.c, "ddpm_conv1x1_fwd", C.byref(d), src.ptr, None,
                     _ptr(w), C.byref(ep), out.ptr, _stream())

    def convT(self, src, key, out):
        """ConvTranspose2d(k=2,s=2) of `src`, padded/cropped (centre, floor-division offsets) into `out`."""
        b, h, w = src.bhw
        _, oh, ow = out.bhw
        dy, dx = oh - 2 * h, ow - 2 * w
        off_y, off_x = dy // 2, dx // 2            # python floor division, also for negatives
        if dy > 0 or dx > 0:                       # positive padding leaves a zero border
            out.buf[..., out.c0:out.c0 + out.c].zero_()
        cout = out.c
        d = ConvDesc(batch=b, in_h=h, in_w=w, c1=src.c, ld1=src.ld, c2=0, ld2=0, cout=cout, ldo=out.ld, act=ACT_NONE,
                     ldr=0, out_h=oh, out_w=ow, off_y=off_y, off_x=off_x)
        ep = Epilogue(None, _ptr(self.T(key + ".bias")), None, None)
        call("ddpm_convT2x2_fwd", C.byref(d), src.ptr, _ptr(self.packed_convT(key)), C.byref(ep), out.ptr, _stream())

    def pool_skip(self, x, temb, skip, pooled):
        b, h, w = x.bhw
        call("ddpm_pool_skip_fwd", x.ptr, x.ld, _ptr(temb), skip.ptr if skip is not None else None,
             skip.ld if skip is not None else 0, pooled.ptr if pooled is not None else None,
             pooled.ld if pooled is not None else 0, None, b, h, w, x.c, _stream())

    def add_temb(self, x, temb, out):
        b, h, w = x.bhw
        call("ddpm_add_temb", x.ptr, x.ld, _ptr(temb), out.ptr, out.ld, b, h * w, x.c, _stream())

    def time_embed(self, t, y, emb_out=None, defer_label_check=False, uniform=False):
        """The four (time [+ label]) embedding vectors per sample. uniform: the caller guarantees that every sample has the
        same timestep (the sampling loop); with at most one label the rows are then equal and computed once."""
        B = t.shape[0]
        dev = t.device
        chans = [self.base, self.base * 2, self.base * 4, self.base * 8]
        outs = [torch.empty((B, c), device=dev, dtype=torch.float32) for c in chans]
        a = EmbedArgs()
        a.batch, a.emb_dim = B, self.emb
        a.uniform = int(bool(uniform) and (y is None or y.numel() == 1))
        if t.dtype == torch.int64:
            a.t_is_float = 0
        elif t.dtype == torch.float32:
            a.t_is_float = 1
        else:
            t = t.to(torch.int64)
            a.t_is_float = 0
        t = t.contiguous()
        a.t = t.data_ptr()
        keep = [t]
        if y is not None:
            if self.num_classes is None:
                raise AttributeError("this UNet was built without num_classes; it has no label embedding")
            y = y.to(device=dev, dtype=torch.int64).contiguous()
            if y.numel() not in (1, B):
                raise RuntimeError(f"label tensor of {y.numel()} entries cannot broadcast over a batch of {B}")
            table = self.T("label_embedding.weight")
            self._check_labels(y, table.shape[0], defer_label_check)
            keep.append(y)
            a.n_labels = y.numel()
            a.n_classes = table.shape[0]
            a.y = y.data_ptr()
            a.label_table = table.data_ptr()
        else:
            a.n_labels = 0
        for k in range(4):
            a.ch[k] = chans[k]
            a.w[k] = self.T(f"time_fc{k + 1}.weight").data_ptr()
            a.b[k] = self.T(f"time_fc{k + 1}.bias").data_ptr()
            if y is not None:
                a.lw[k] = self.T(f"label_emdedding_fc{k + 1}.weight").data_ptr()
                a.lb[k] = self.T(f"label_emdedding_fc{k + 1}.bias").data_ptr()
            a.out[k] = outs[k].data_ptr()
        a.emb_out = emb_out.data_ptr() if emb_out is not None else None
        call("ddpm_time_embed", C.byref(a), _stream())
        return outs

    def _check_labels(self, y, n_classes, defer=False):
        """nn.Embedding raises for a label outside [0, num_classes) (third-iteration/unet.py:136); so does this, with
        one device->host read per DISTINCT label tensor: the sampling loop hands the same tensor to every one of its
        T steps and pays once. (The kernels never dereference such a label either way: that sample's embedding rows
        are NaN.) A training step brings a fresh label tensor every time, and a read there would stall the host at
        the start of every forward: with `defer` the verdict is copied to pinned memory behind the step's kernels and
        examined, without waiting, at the start of a later call, which then raises."""
        seen = getattr(self, "_labels_ok", None)      # (weak reference to the tensor object, its version, class count)
        if seen is not None and seen[0]() is y and seen[1:] == (y._version, n_classes):
            return
        if defer:
            bad = ((y < 0) | (y >= n_classes)).any().to(torch.int32).view(1)
            host = torch.empty(1, dtype=torch.int32, pin_memory=True)
            host.copy_(bad, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._labels_pending.append((ev, host, n_classes))
            return
        lo, hi = torch.aminmax(y)
        if int(lo) < 0 or int(hi) >= n_classes:
            raise IndexError(f"label out of range: labels span [{int(lo)}, {int(hi)}], the embedding has {n_classes} classes")
        self._labels_ok = (weakref.ref(y), y._version, n_classes)

    # ------------------------------------------------------------------ blocks (eval mode)
    def _act(self):
        return ACT_RELU if self.v == 1 else ACT_GELU

    def double_conv_eval(self, src, key, cout, out, temb=None, first=None):
        """DoubleConv with BatchNorm folded into the conv epilogues. `first` = NCHW image for enc1."""
        p = key + ".double_conv"
        b, h, w = out.bhw
        act = self._act()
        s0, h0 = self.folded_bn(p + ".0", p + ".1")
        a = View(self._new(b, h, w, cout, out.buf))
        if first is not None:
            ep = Epilogue(_ptr(s0), _ptr(h0), None, None)
            wt = self.T(p + ".0.weight")
            call("ddpm_conv3x3_first_fwd", _ptr(first), _ptr(wt), C.byref(ep), act, a.ptr, b, first.shape[1], h, w, cout,
                 a.ld, _stream())
        else:
            self.conv3x3(src, self.packed_conv3(p + ".0"), cout, a, s0, h0, act=act)
        if self.v == 1:
            s1, h1 = self.folded_bn(p + ".3", p + ".4")
            self.conv3x3(a, self.packed_conv3(p + ".3"), cout, out, s1, h1, temb=temb, act=act)
            return
        r = p + ".3.conv_block"
        s1, h1 = self.folded_bn(r + ".0", r + ".1")
        s2, h2 = self.folded_bn(r + ".3", r + ".4")
        m = View(self._new(b, h, w, cout, out.buf))
        self.conv3x3(a, self.packed_conv3(r + ".0"), cout, m, s1, h1, act=act)
        self.conv3x3(m, self.packed_conv3(r + ".3"), cout, out, s2, h2, residual=a, temb=temb, act=act)

    def attention(self, x, key, out, temb=None, save=None):
        b, h, w = x.bhw
        c = x.c
        wq, bq, dqp, _ = self.packed_qkv(key)
        if dqp not in (8, 16, 32, 64, 128):
            raise NotImplementedError(f"attention with C/8 = {c // 8} query channels is not supported (at most 128: base_channels 128)")
        qkv = View(self._new(b, h, w, 2 * dqp + c, x.buf))
        n = h * w
        # mirrors the library's choice of kernel family (profiling label only)
        math = conv_math()
        tag = "fp32"
        if n >= 128 and c >= 64 and math != "fp32" and dqp <= 64:
            pair = math == "f16x2" and dqp <= 16 and (save is None or os.environ.get("DDPM_ATTN_TRAIN_F16", "1") != "0")
            tag = "fp16-pair" if pair else "split-bf16"
        if save is None and _FUSED_ATTENTION:
            # inference: projection and attention behind one entry point, which may hand k and v over as fp16 pairs
            self._launch(f"self_attention<inference,{tag}>", 2.0 * b * n * (2 * dqp + c) * c + 2.0 * b * n * n * (dqp + c),
                         "ddpm_self_attention_infer", x.ptr, x.ld, _ptr(wq), _ptr(bq), _ptr(self.T(key + ".gamma")), _ptr(temb),
                         out.ptr, out.ld, qkv.ptr, b, n, c, dqp, _stream())
            return
        self.conv1x1(x, wq, bq, 2 * dqp + c, qkv)
        raw = lse = None
        if save is not None:
            raw = torch.empty((b, h, w, c), device=x.buf.device, dtype=torch.float32)
            lse = torch.empty((b, h * w), device=x.buf.device, dtype=torch.float32)
            save.update(qkv=qkv, raw=raw, lse=lse)
        self._launch(f"attention_fwd<{tag}>", 2.0 * b * n * n * (dqp + c), "ddpm_attention_fwd", qkv.ptr, x.ptr, x.ld,
                     _ptr(self.T(key + ".gamma")), out.ptr, out.ld, _ptr(raw), _ptr(lse), _ptr(temb), b, n, c, dqp, _stream())

    # ------------------------------------------------------------------ whole network, eval mode
    def forward_eval(self, x, t, y=None, want_fms=True, step=None):
        """Eval-mode forward. With `step` (a dict: c1, c2, c3, noise_mode, z, seed, call, eps_cond, cfg_w, out, t_uniform)
        the last layer and the reverse ancestral step run as one kernel and the return value is x_{t-1} instead of
        eps-hat; t_uniform says that all entries of t are equal (the embedding is then computed once)."""
        B, cin, H, W = x.shape
        if cin != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {cin}")
        x = x.contiguous()
        dev = x.device
        bc = self.base
        v1 = self.v == 1
        temb = self.time_embed(t, y, uniform=bool(step is not None and step.get("t_uniform")))
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        fms = []

        def buf(h, w, c):
            return torch.empty((B, h, w, c), device=dev, dtype=torch.float32)

        # concatenation buffers [skip | upsampled] of the three UpBlocks
        cat3, cat2, cat1 = buf(H, W, 2 * bc), buf(H, W, 4 * bc), buf(H2, W2, 8 * bc)
        skip1, skip2, skip3 = View(cat3, 0, bc), View(cat2, 0, 2 * bc), View(cat1, 0, 4 * bc)

        # enc1: the skip (with embedding) is also the input of enc2
        self.double_conv_eval(None, "enc1", bc, skip1, temb=temb[0], first=x)
        fms.append(skip1)

        def down(src, key, cout, skip, te, h, w):
            xfull = View(buf(h, w, cout))
            self.double_conv_eval(src, key + ".conv", cout, xfull)
            if not v1:
                xa = View(buf(h, w, cout))
                self.attention(xfull, key + ".attention", xa)
                xfull = xa
            pooled = View(buf(h // 2, w // 2, cout))
            self.pool_skip(xfull, te, skip, pooled)      # pooled copy carries no embedding
            return pooled

        p2 = down(skip1, "enc2", 2 * bc, skip2, temb[1], H, W)
        fms.append(skip2)
        p3 = down(p2, "enc3", 4 * bc, skip3, temb[2], H2, W2)
        fms.append(skip3)

        def tail(src, key_conv, key_attn, cout, te, h, w):
            """DoubleConv (+attention) + embedding add; returns the tensor handed to the next block."""
            if v1 and not want_fms:
                # nobody looks at the feature map (a sampling step): the embedding is added in the conv's epilogue,
                # the same fp32 addition on the same values as the separate kernel below
                post = View(buf(h, w, cout))
                self.double_conv_eval(src, key_conv, cout, post, temb=te)
                return post
            if v1:
                pre = View(buf(h, w, cout))
                self.double_conv_eval(src, key_conv, cout, pre)
                fms.append(pre)                           # v1 records before the embedding add
                post = View(buf(h, w, cout))
                self.add_temb(pre, te, post)
                return post
            mid = View(buf(h, w, cout))
            self.double_conv_eval(src, key_conv, cout, mid)
            post = View(buf(h, w, cout))
            self.attention(mid, key_attn, post, temb=te)
            fms.append(post)
            return post

        hcur = tail(p3, "bottleneck" if v1 else "bottleneck.0", "bottleneck.1", 8 * bc, temb[3], H4, W4)
        for key, cat, cskip, cout, te, (h, w) in (("dec1", cat1, 4 * bc, 4 * bc, temb[2], (H2, W2)),
                                                   ("dec2", cat2, 2 * bc, 2 * bc, temb[1], (H, W)),
                                                   ("dec3", cat3, bc, bc, temb[0], (H, W))):
            self.convT(hcur, key + ".up", View(cat, cskip, cskip))
            hcur = tail(View(cat), key + ".conv", key + ".attention", cout, te, h, w)

        fw = self.T("final_conv.weight")
        if step is not None:
            if self.cout_final != cin:
                raise RuntimeError("the fused reverse step needs out_channels == in_channels (eps-hat has the shape of x_t)")
            out = step.get("out")
            if out is None:
                out = torch.empty_like(x)
            z, ec = step.get("z"), step.get("eps_cond")
            call("ddpm_final_conv_reverse_step", hcur.ptr, hcur.ld, _ptr(fw), _ptr(self.T("final_conv.bias")), _ptr(x), _ptr(ec),
                 float(step.get("cfg_w", 0.0)), int(step["noise_mode"]), _ptr(z), C.c_uint64(step.get("seed", 0)),
                 C.c_uint64(step.get("call", 0)), step["c1"], step["c2"], step["c3"], _ptr(out), _ptr(step.get("eps_out")), B, H, W, bc,
                 self.cout_final, _stream())
            return out, ([f.nchw() for f in fms] if want_fms else [])
        out = torch.empty((B, self.cout_final, H, W), device=dev, dtype=torch.float32)
        call("ddpm_conv1x1_final_fwd", hcur.ptr, hcur.ld, _ptr(fw), _ptr(self.T("final_conv.bias")), _ptr(out), B, H, W, bc,
             self.cout_final, _stream())
        return out, ([f.nchw() for f in fms] if want_fms else [])
