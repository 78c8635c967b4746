"""Train-mode forward (batch-statistics BatchNorm, saved activations) and the hand-written backward
schedule of the UNet, as calls into libddpm_mi355x.so. Mixed into UNetEngine.

The whole network is ONE autograd node (UNetFunction): forward records what the backward needs in
a plain Python structure, backward walks it in reverse and returns one gradient per parameter, so
`loss.backward()` / `optim.Adam.step()` of the reference's training loop
(first-iteration/main.py:76-84) work unchanged.
"""
import contextlib
import ctypes as C
import os

import torch

from ._lib import ConvDesc, EmbedArgs, EmbedBwdArgs, Epilogue, call, conv_math, lib, packed_floats

ACT_NONE, ACT_RELU, ACT_GELU = 0, 1, 2
BN_EPS, BN_MOMENTUM = 1e-5, 0.1


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _dist_active():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


class TrainMixin:
    sync_bn = False     # data-parallel training: BatchNorm over the GLOBAL batch (statistics all-reduced across ranks;
                        # every rank must hold the same number of samples: the global row count is local rows x world size)
    dist_group = None   # process group of those all-reduces (None = the default group); BucketedGradientAverager.attach sets it
    grad_sink = None    # data-parallel training: where the backward puts parameter gradients (parallel.BucketedGradientAverager)
    _s2 = None                                                      # second stream of the backward pass (weight gradients)
    _main_stream = None                                             # the stream the backward itself runs on (set when _s2 is used)
    _use_side = os.environ.get("DDPM_SIDE_STREAM", "1") != "0"      # diagnostic switch: "0" keeps the backward on one stream
    _fused_stats = os.environ.get("DDPM_BN_FUSED", "1") != "0"     # BatchNorm statistics summed in the 3x3 convolution's epilogue (no separate pass over its output); "0": diagnostic switch
    _gall = ()          # gradient tensors this backward allocated itself (not slices of an averager's buffer)
    _gstate = None      # three device floats: largest |incoming gradient|, the backward's power-of-two scale, its inverse

    _shards_checked = False

    def _check_equal_shards(self, batch):
        """Global-batch BatchNorm and the gradient mean over ranks both assume that every rank holds the same number of
        samples. Checked once, at this network's first training step under sync_bn (which all ranks reach together):
        a global batch that does not divide by the world size fails here instead of training on skewed statistics."""
        import torch.distributed as dist
        sizes = [None] * dist.get_world_size(self.dist_group)
        dist.all_gather_object(sizes, int(batch), group=self.dist_group)
        self._shards_checked = True
        if len(set(sizes)) != 1:
            raise ValueError(f"sync_bn needs the same number of samples on every rank, got {sizes}: cut the global batch evenly "
                             "(or train with sync_bn = False, per-rank statistics)")

    # ------------------------------------------------------------------ scratch
    def _workspace(self, dev):
        ws = getattr(self, "_ws", None)
        if ws is None or ws.device != dev:
            n = lib().ddpm_reduce_workspace_doubles(max(8 * self.base, 64))
            self._ws = torch.empty(n, dtype=torch.float64, device=dev)
            self._sums = torch.empty(2 * 8 * self.base, dtype=torch.float32, device=dev)
        return self._ws

    @staticmethod
    def _wgrad_label(kind, m, n):
        """Kernel family the library uses for a weight gradient (profiling label): tiles of 128 run on split operands."""
        math = conv_math()
        split = (m > 64 or n > 64 or (kind == "conv3x3" and 32 < m <= 64 and 32 < n <= 64)) and math != "fp32"
        return f"conv_wgrad<{kind},{('split-bf16' if math == 'bf16x3' else 'fp16-pair') if split else 'fp32'}>"

    # ------------------------------------------------------------------ gradient storage
    def _gnew(self, shape, dev, zero=False):
        """Storage of one parameter gradient (or of a fused group of them): a fresh tensor, or, in data-parallel
        training, the next slice of the averager's flat buffer, so that buckets of finished gradients can be
        all-reduced while the rest of the backward is still running and nothing is copied afterwards."""
        shape = tuple(shape) if isinstance(shape, (tuple, list, torch.Size)) else (int(shape),)
        sink = self.grad_sink
        if sink is None:
            t = (torch.zeros if zero else torch.empty)(shape, device=dev, dtype=torch.float32)
            self._gall.append(t)
            return t
        n = 1
        for s in shape:
            n *= s
        t = sink.alloc(n, dev).view(shape)
        if not sink.owns(t):
            self._gall.append(t)
        if zero:
            t.zero_()
        return t

    def _gflush(self):
        """A point of the backward at which every gradient handed out so far has all its kernels enqueued."""
        if self.grad_sink is not None:
            self.grad_sink.flush(self)

    # ------------------------------------------------------------------ forward pieces
    def _conv_bn_fwd(self, V, src, conv_key, bn_key, cout, act, out, residual=None, temb=None, first=None):
        """conv3x3 (+bias) -> batch statistics -> normalise/affine (+residual) -> act (+temb)."""
        b, h, w = out.bhw
        dev = out.buf.device
        raw = V(torch.empty((b, h, w, cout), device=dev, dtype=torch.float32))
        bias = self.T(conv_key + ".bias")
        mean = torch.empty(cout, device=dev, dtype=torch.float32)
        invstd = torch.empty(cout, device=dev, dtype=torch.float32)
        stat_rows = b * h * w
        global_stats = self.sync_bn and _dist_active()
        fused = self._fused_stats and first is None
        if first is not None:
            ep = Epilogue(None, _ptr(bias), None, None)
            call("ddpm_conv3x3_first_fwd", _ptr(first), _ptr(self.T(conv_key + ".weight")), C.byref(ep), ACT_NONE, raw.ptr, b,
                 first.shape[1], h, w, cout, raw.ld, _stream())
        elif fused:
            # the convolution's epilogue leaves the partial sums of the batch statistics: no separate pass over `raw`
            nblk = self.conv3x3_stats(src, self.packed_conv3(conv_key), cout, raw, bias, self._workspace(dev))
            if not global_stats:
                call("ddpm_bn_stats_from_partials", _ptr(self._workspace(dev)), nblk, stat_rows, cout, BN_EPS, BN_MOMENTUM, _ptr(mean),
                     _ptr(invstd), _ptr(self.T(bn_key + ".running_mean")), _ptr(self.T(bn_key + ".running_var")), _stream())
        else:
            self.conv3x3(src, self.packed_conv3(conv_key), cout, raw, None, bias)
        if fused and not global_stats:
            pass
        elif global_stats:
            # global-batch statistics: this rank's sums are all-reduced, then finalised on every rank. Every rank
            # holds the same number of samples (the batch axis is cut evenly), so the global row count needs no exchange.
            from . import parallel
            sums = torch.empty(2 * cout, device=dev, dtype=torch.float64)
            if fused:
                call("ddpm_bn_sums_from_partials", _ptr(self._workspace(dev)), nblk, cout, _ptr(sums), _stream())
            else:
                call("ddpm_bn_partial_sums", raw.ptr, raw.ld, stat_rows, cout, _ptr(self._workspace(dev)), _ptr(sums), _stream())
            parallel.all_reduce_(sums, group=self.dist_group)
            stat_rows = stat_rows * parallel.world_size(self.dist_group)
            call("ddpm_bn_finalize_stats", _ptr(sums), stat_rows, cout, BN_EPS, BN_MOMENTUM, _ptr(mean), _ptr(invstd),
                 _ptr(self.T(bn_key + ".running_mean")), _ptr(self.T(bn_key + ".running_var")), _stream())
        else:
            call("ddpm_bn_stats", raw.ptr, raw.ld, stat_rows, cout, BN_EPS, BN_MOMENTUM, _ptr(self._workspace(dev)), _ptr(mean),
                 _ptr(invstd), _ptr(self.T(bn_key + ".running_mean")), _ptr(self.T(bn_key + ".running_var")), _stream())
        self._nbt.append(self.T(bn_key + ".num_batches_tracked"))    # incremented together at the end of the forward
        self._stats_written += [self.T(bn_key + ".running_mean"), self.T(bn_key + ".running_var")]
        call("ddpm_bn_apply_fwd", raw.ptr, raw.ld, _ptr(mean), _ptr(invstd), _ptr(self.T(bn_key + ".weight")),
             _ptr(self.T(bn_key + ".bias")), residual.ptr if residual is not None else None,
             residual.ld if residual is not None else 0, _ptr(temb), act, out.ptr, out.ld, b, h * w, cout, _stream())
        return dict(kind="conv_bn", src=src, raw=raw, mean=mean, invstd=invstd, conv=conv_key, bn=bn_key, act=act,
                    residual=residual, first=first, cout=cout, stat_rows=stat_rows)

    def _double_conv_fwd(self, V, src, key, cout, out, temb=None, first=None):
        p = key + ".double_conv"
        b, h, w = out.bhw
        dev = out.buf.device
        act = ACT_RELU if self.v == 1 else ACT_GELU
        new = lambda: V(torch.empty((b, h, w, cout), device=dev, dtype=torch.float32))
        a = new()
        recs = [self._conv_bn_fwd(V, src, p + ".0", p + ".1", cout, act, a, first=first)]
        if self.v == 1:
            recs.append(self._conv_bn_fwd(V, a, p + ".3", p + ".4", cout, act, out, temb=temb))
        else:
            r = p + ".3.conv_block"
            m = new()
            recs.append(self._conv_bn_fwd(V, a, r + ".0", r + ".1", cout, act, m))
            recs.append(self._conv_bn_fwd(V, m, r + ".3", r + ".4", cout, act, out, residual=a, temb=temb))
        return recs

    def _attention_fwd_train(self, V, x, key, out, temb=None):
        save = {}
        self.attention(x, key, out, temb=temb, save=save)
        return dict(kind="attn", x=x, key=key, qkv=save["qkv"], raw=save["raw"], lse=save["lse"])

    # ------------------------------------------------------------------ whole network, train mode
    def forward_train(self, x, t, y=None):
        from .engine import View as V
        B, cin, H, W = x.shape
        if cin != self.cin:
            raise RuntimeError(f"expected {self.cin} input channels, got {cin}")
        x = x.contiguous()
        dev = x.device
        if self.sync_bn and not self._shards_checked and _dist_active():
            self._check_equal_shards(B)
        self.prepack_conv3()          # every 3x3 weight the optimizer has changed, both orientations, in one launch
        bc, v1 = self.base, self.v == 1
        emb = torch.empty((B, self.emb // 2 * 2), device=dev, dtype=torch.float32)
        temb = self.time_embed(t, y, emb_out=emb, defer_label_check=True)   # no host wait inside a training step
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        fms = []
        tape = dict(x=x, t=t, y=y, emb=emb, shape=(B, H, W))
        self._nbt, self._stats_written = [], []

        def buf(h, w, c):
            return torch.empty((B, h, w, c), device=dev, dtype=torch.float32)

        cat3, cat2, cat1 = buf(H, W, 2 * bc), buf(H, W, 4 * bc), buf(H2, W2, 8 * bc)
        skip1, skip2, skip3 = V(cat3, 0, bc), V(cat2, 0, 2 * bc), V(cat1, 0, 4 * bc)
        tape["enc1"] = self._double_conv_fwd(V, None, "enc1", bc, skip1, temb=temb[0], first=x)
        fms.append(skip1)

        def down(src, key, cout, skip, te, h, w):
            rec = {}
            xfull = V(buf(h, w, cout))
            rec["dc"] = self._double_conv_fwd(V, src, key + ".conv", cout, xfull)
            if not v1:
                xa = V(buf(h, w, cout))
                rec["attn"] = self._attention_fwd_train(V, xfull, key + ".attention", xa)
                xfull = xa
            pooled = V(buf(h // 2, w // 2, cout))
            amax = torch.empty((B, h // 2, w // 2, cout), device=dev, dtype=torch.uint8)
            call("ddpm_pool_skip_fwd", xfull.ptr, xfull.ld, _ptr(te), skip.ptr, skip.ld, pooled.ptr, pooled.ld, _ptr(amax), B, h, w,
                 cout, _stream())
            rec.update(argmax=amax, hw=(h, w), cout=cout)
            return pooled, rec

        p2, tape["enc2"] = down(skip1, "enc2", 2 * bc, skip2, temb[1], H, W)
        fms.append(skip2)
        p3, tape["enc3"] = down(p2, "enc3", 4 * bc, skip3, temb[2], H2, W2)
        fms.append(skip3)

        def tail(src, key_conv, key_attn, cout, te, h, w):
            rec = {}
            if v1:
                pre = V(buf(h, w, cout))
                rec["dc"] = self._double_conv_fwd(V, src, key_conv, cout, pre)
                fms.append(pre)
                post = V(buf(h, w, cout))
                self.add_temb(pre, te, post)
                return post, rec
            mid = V(buf(h, w, cout))
            rec["dc"] = self._double_conv_fwd(V, src, key_conv, cout, mid)
            post = V(buf(h, w, cout))
            rec["attn"] = self._attention_fwd_train(V, mid, key_attn, post, temb=te)
            fms.append(post)
            return post, rec

        hcur, tape["bottleneck"] = tail(p3, "bottleneck" if v1 else "bottleneck.0", "bottleneck.1", 8 * bc, temb[3], H4, W4)
        for key, cat, cskip, cout, te, (h, w) in (("dec1", cat1, 4 * bc, 4 * bc, temb[2], (H2, W2)),
                                                   ("dec2", cat2, 2 * bc, 2 * bc, temb[1], (H, W)),
                                                   ("dec3", cat3, bc, bc, temb[0], (H, W))):
            up_out = V(cat, cskip, cskip)
            self.convT(hcur, key + ".up", up_out)
            up_in = hcur
            hcur, rec = tail(V(cat), key + ".conv", key + ".attention", cout, te, h, w)
            rec.update(up_in=up_in, up_out=up_out, cskip=cskip)
            tape[key] = rec

        out = torch.empty((B, self.cout_final, H, W), device=dev, dtype=torch.float32)
        call("ddpm_conv1x1_final_fwd", hcur.ptr, hcur.ld, _ptr(self.T("final_conv.weight")), _ptr(self.T("final_conv.bias")),
             _ptr(out), B, H, W, bc, self.cout_final, _stream())
        tape["final_in"] = hcur
        torch._foreach_add_(self._nbt, 1)
        # the kernels updated the running statistics through raw pointers: tell torch (and the eval-mode cache of
        # folded scale/shift pairs, which is keyed on tensor versions) that these buffers changed
        torch.autograd.graph.increment_version(self._stats_written)
        self._nbt, self._stats_written = [], []
        return out, [f.nchw() for f in fms], tape

    # ------------------------------------------------------------------ backward pieces
    def _side(self, *tensors):
        """Context in which launches go to a second stream that first waits for everything issued so far on
        the current one. Weight gradients have no consumer before the optimizer step, so they run there,
        beside the chain of data gradients (which is a string of short reduction/packing kernels between
        the large ones). `tensors` are marked as used by that stream, so that the caching allocator does not
        hand their memory out again before the side stream is done with it. backward() joins the streams."""
        if not self._use_side:
            return contextlib.nullcontext()
        main = torch.cuda.current_stream()
        if self._s2 is None:
            self._s2 = torch.cuda.Stream(device=main.device)
        self._main_stream = main
        self._s2.wait_stream(main)
        for t in tensors:
            t.record_stream(self._s2)
        return torch.cuda.stream(self._s2)

    def _conv_bn_bwd(self, V, rec, dout, grads, add=None, need_dsrc=True):
        """dout: View, gradient of the block's output. Returns the gradient w.r.t. rec['src'] (a View)
        and the gradient w.r.t. the residual input (a View or None)."""
        raw, cout = rec["raw"], rec["cout"]
        b, h, w = raw.bhw
        dev = raw.buf.device
        rows = b * h * w
        ws = self._workspace(dev)
        draw = V(torch.empty((b, h, w, cout), device=dev, dtype=torch.float32))
        res = rec["residual"]
        dres = V(torch.empty((b, h, w, cout), device=dev, dtype=torch.float32)) if res is not None else None
        bn, conv = rec["bn"], rec["conv"]
        dw, db, dcb = (self._gnew(cout, dev) for _ in range(3))
        rp, rl = (res.ptr, res.ld) if res is not None else (None, 0)
        dp, dl = (dres.ptr, dres.ld) if dres is not None else (None, 0)
        if rec["stat_rows"] != rows:
            # statistics were taken over all ranks' rows: the two sums of the backward are all-reduced as well
            from . import parallel
            sums = torch.empty(2 * cout, device=dev, dtype=torch.float64)
            call("ddpm_bn_bwd_sums", raw.ptr, raw.ld, dout.ptr, dout.ld, rp, rl, _ptr(rec["mean"]), _ptr(rec["invstd"]),
                 _ptr(self.T(bn + ".weight")), _ptr(self.T(bn + ".bias")), rec["act"], rows, cout, _ptr(ws), _ptr(sums), _stream())
            db.copy_(sums[:cout])                              # this rank's share of the parameter gradients
            dw.copy_(sums[cout:])
            parallel.all_reduce_(sums, group=self.dist_group)
            call("ddpm_bn_bwd_apply", raw.ptr, raw.ld, dout.ptr, dout.ld, rp, rl, _ptr(rec["mean"]), _ptr(rec["invstd"]),
                 _ptr(self.T(bn + ".weight")), _ptr(self.T(bn + ".bias")), _ptr(sums.float()), rec["stat_rows"], rec["act"], rows, cout,
                 _ptr(ws), draw.ptr, draw.ld, dp, dl, _ptr(dcb), _stream())
        else:
            call("ddpm_bn_bwd", raw.ptr, raw.ld, dout.ptr, dout.ld, rp, rl, _ptr(rec["mean"]), _ptr(rec["invstd"]),
                 _ptr(self.T(bn + ".weight")), _ptr(self.T(bn + ".bias")), rec["act"], rows, cout, _ptr(ws), _ptr(self._sums), _ptr(dw),
                 _ptr(db), 0, draw.ptr, draw.ld, dp, dl, _ptr(dcb), _stream())
        grads[bn + ".weight"], grads[bn + ".bias"], grads[conv + ".bias"] = dw, db, dcb
        wt = self.T(conv + ".weight")
        gw = self._gnew(wt.shape, dev)
        if rec["first"] is not None:
            img = rec["first"]
            call("ddpm_conv3x3_first_wgrad", draw.ptr, draw.ld, _ptr(img), _ptr(ws), _ptr(gw), b, img.shape[1], h, w, cout, _stream())
            grads[conv + ".weight"] = gw
            self._gflush()
            return None, dres
        src = rec["src"]
        packed = torch.empty(wt.numel(), device=dev, dtype=torch.float32)
        with self._side(draw.buf, packed, gw):   # the weight gradient runs beside the data-gradient chain
            self._launch(self._wgrad_label("conv3x3", cout, src.c), 2.0 * rows * cout * 9 * src.c, "ddpm_conv3x3_wgrad", draw.ptr, draw.ld, cout, src.ptr,
                         src.ld, src.c, b, h, w, _ptr(packed), _ptr(gw), _stream())
        grads[conv + ".weight"] = gw
        self._gflush()
        dsrc = None
        if need_dsrc:
            dsrc = V(torch.empty((b, h, w, src.c), device=dev, dtype=torch.float32))
            self.conv3x3(draw, self.packed_conv3(conv, flip=True), src.c, dsrc, residual=add)
        return dsrc, dres

    def _double_conv_bwd(self, V, recs, dout, grads, add=None, need_dsrc=True):
        if self.v == 1:
            d_a, _ = self._conv_bn_bwd(V, recs[1], dout, grads)
            return self._conv_bn_bwd(V, recs[0], d_a, grads, add=add, need_dsrc=need_dsrc)[0]
        d_m, d_a_res = self._conv_bn_bwd(V, recs[2], dout, grads)
        d_a, _ = self._conv_bn_bwd(V, recs[1], d_m, grads, add=d_a_res)
        return self._conv_bn_bwd(V, recs[0], d_a, grads, add=add, need_dsrc=need_dsrc)[0]

    def _attention_bwd(self, V, rec, dout, grads):
        x, key, qkv = rec["x"], rec["key"], rec["qkv"]
        b, h, w = x.bhw
        c, n = x.c, h * w
        dev = x.buf.device
        _, bq, dqp, wt = self.packed_qkv(key)
        lq = 2 * dqp + c
        dq = c // 8
        dqkv = V(torch.empty((b, h, w, lq), device=dev, dtype=torch.float32))
        dgamma = self._gnew(1, dev)
        e = torch.empty((b, n), device=dev, dtype=torch.float32)
        ws = self._workspace(dev)
        x3 = conv_math() != "fp32" and (c, dqp) in ((64, 8), (128, 16), (256, 32))
        pair = x3 and conv_math() == "f16x2" and dqp <= 16 and os.environ.get("DDPM_ATTN_TRAIN_F16", "1") != "0"
        self._launch(("attention_bwd<fp16-pair>" if pair else "attention_bwd<split-bf16>") if x3 else "attention_bwd<fp32>", 2.0 * b * n * n * (2 * dqp + 2 * c + dqp + c), "ddpm_attention_bwd", qkv.ptr, dout.ptr, dout.ld,
                     _ptr(rec["raw"]), _ptr(rec["lse"]), _ptr(self.T(key + ".gamma")), _ptr(ws), _ptr(e), dqkv.ptr, _ptr(dgamma), b, n, c,
                     dqp, _stream())
        grads[key + ".gamma"] = dgamma
        gw = self._gnew((lq, c), dev)                        # query | key | value weight gradients, one fused slice
        with self._side(dqkv.buf, gw):
            self._launch(self._wgrad_label("conv1x1", lq, c), 2.0 * b * n * lq * c, "ddpm_conv1x1_wgrad", dqkv.ptr, lq, lq, x.ptr, x.ld, c, b * n,
                         _ptr(gw), _stream())
        gb = self._gnew(lq, dev)
        for c0 in range(0, lq, 1024):                        # the column-sum kernel takes at most 1024 columns a call
            cw = min(1024, lq - c0)
            call("ddpm_colsum", C.c_void_p(dqkv.buf.data_ptr() + 4 * c0), lq, 1, b * n, cw, _ptr(ws), 0, C.c_void_p(gb.data_ptr() + 4 * c0),
                 _stream())
        self._gflush()
        for name, lo, cnt in (("query", 0, dq), ("key", dqp, dq), ("value", 2 * dqp, c)):
            grads[f"{key}.{name}.weight"] = gw[lo:lo + cnt].reshape(self.T(f"{key}.{name}.weight").shape)
            grads[f"{key}.{name}.bias"] = gb[lo:lo + cnt]
        # dx = dqkv @ Wqkv + dout   (1x1 conv with the transposed weight, residual = pass-through gradient)
        dx = V(torch.empty((b, h, w, c), device=dev, dtype=torch.float32))
        d = ConvDesc(batch=b, in_h=h, in_w=w, c1=lq, ld1=lq, c2=0, ld2=0, cout=c, ldo=c, act=ACT_NONE, ldr=dout.ld, out_h=h, out_w=w,
                     off_y=0, off_x=0)
        ep = Epilogue(None, None, dout.ptr, None)
        self._launch(self._gemm_label("conv1x1", b * n, c), 2.0 * b * n * c * lq, "ddpm_conv1x1_fwd", C.byref(d), dqkv.ptr, None,
                     _ptr(wt), C.byref(ep), dx.ptr, _stream())
        return dx

    def _temb_grad(self, dview, dtemb, k, dev):
        """dtemb[k] (+)= per-sample column sums of the gradient at a `+ time_emb` site."""
        b, h, w = dview.bhw
        first = dtemb[k] is None
        if first:
            dtemb[k] = torch.empty((b, dview.c), device=dev, dtype=torch.float32)
        call("ddpm_colsum", dview.ptr, dview.ld, b, h * w, dview.c, _ptr(self._workspace(dev)), 0 if first else 1, _ptr(dtemb[k]),
             _stream())

    # ------------------------------------------------------------------ whole network backward
    def backward(self, tape, dout):
        from .engine import View as V
        B, H, W = tape["shape"]
        dev = dout.device
        bc, v1 = self.base, self.v == 1
        grads, dtemb = {}, [None, None, None, None]
        ws = self._workspace(dev)
        dout = dout.contiguous()
        self._gall = []
        # The backward runs on the incoming gradient times the power of two that brings its largest magnitude into
        # [1, 2) (loss gradients are far below fp16's exponent range, which the fp16-pair arithmetic needs its operands
        # in), and every parameter gradient is multiplied by the inverse at the end; both are exact in fp32. The
        # three-plane bf16 arithmetic does not need it but does the same, so that the ranks of a data-parallel job issue
        # the same collectives whichever of the two each of them is in. Ranks agree on the scale first: their
        # gradients (and global-batch BatchNorm sums) are added to each other.
        scaled = conv_math() != "fp32"
        if scaled:
            if self._gstate is None or self._gstate.device != dev:
                self._gstate = torch.zeros(3, device=dev, dtype=torch.float32)
            st = self._gstate
            call("ddpm_grad_amax", _ptr(dout), dout.numel(), _ptr(st), _stream())
            if _dist_active():
                import torch.distributed as dist
                from . import parallel
                parallel.all_reduce_(st[0:1], op=dist.ReduceOp.MAX, group=self.dist_group)
            d2 = torch.empty_like(dout)
            call("ddpm_grad_scale", _ptr(dout), dout.numel(), _ptr(st), _ptr(d2), _stream())
            dout = d2

        # final 1x1 conv
        hfin = tape["final_in"]
        dh = V(torch.empty((B, H, W, bc), device=dev, dtype=torch.float32))
        gw = self._gnew(self.T("final_conv.weight").shape, dev)
        gb = self._gnew(self.T("final_conv.bias").shape, dev)
        call("ddpm_conv1x1_final_bwd", _ptr(dout), hfin.ptr, hfin.ld, _ptr(self.T("final_conv.weight")), _ptr(ws), dh.ptr, dh.ld,
             _ptr(gw), _ptr(gb), B, H, W, bc, self.cout_final, _stream())
        grads["final_conv.weight"], grads["final_conv.bias"] = gw, gb

        def tail_bwd(rec, dpost, k, add=None, need_dsrc=True):
            self._temb_grad(dpost, dtemb, k, dev)
            d = dpost
            if not v1:
                d = self._attention_bwd(V, rec["attn"], dpost, grads)
            return self._double_conv_bwd(V, rec["dc"], d, grads, add=add, need_dsrc=need_dsrc)

        dskips = {}
        for key, k in (("dec3", 0), ("dec2", 1), ("dec1", 2)):
            rec = tape[key]
            dcat = tail_bwd(rec, dh, k)                                   # [B,h,w,2*cskip]
            cs = rec["cskip"]
            dskips[key] = V(dcat.buf, 0, cs)
            dup = V(dcat.buf, cs, cs)
            up_in, up_out = rec["up_in"], rec["up_out"]
            b, h, w = up_in.bhw
            _, oh, ow = up_out.bhw
            off_y, off_x = (oh - 2 * h) // 2, (ow - 2 * w) // 2
            ukey = key + ".up"
            wt = self.T(ukey + ".weight")
            guw, packed = self._gnew(wt.shape, dev), torch.empty(wt.numel(), device=dev, dtype=torch.float32)
            with self._side(dcat.buf, packed, guw):
                call("ddpm_convT2x2_wgrad", dup.ptr, dup.ld, cs, oh, ow, off_y, off_x, up_in.ptr, up_in.ld, up_in.c, b, h, w, _ptr(packed),
                     _ptr(guw), _stream())
            y0, x0 = max(off_y, 0), max(off_x, 0)
            y1, x1 = min(oh, off_y + 2 * h), min(ow, off_x + 2 * w)
            gub = self._gnew(cs, dev)
            call("ddpm_colsum_window", dup.ptr, dup.ld, b, oh, ow, y0, x0, y1 - y0, x1 - x0, cs, _ptr(ws), _ptr(gub), _stream())
            grads[ukey + ".weight"], grads[ukey + ".bias"] = guw, gub
            self._gflush()
            dh = V(torch.empty((b, h, w, up_in.c), device=dev, dtype=torch.float32))
            d = ConvDesc(batch=b, in_h=h, in_w=w, c1=cs, ld1=dup.ld, c2=0, ld2=0, cout=up_in.c, ldo=dh.ld, act=ACT_NONE, ldr=0,
                         out_h=oh, out_w=ow, off_y=off_y, off_x=off_x)
            ep = Epilogue(None, None, None, None)
            wpk = self._cached("pkTd", (ukey + ".weight",), lambda wt_: self._pack_convT_dgrad(wt_))
            call("ddpm_convT2x2_dgrad", C.byref(d), dup.ptr, _ptr(wpk), C.byref(ep), dh.ptr, _stream())

        dp3 = tail_bwd(tape["bottleneck"], dh, 3)                          # gradient of enc3's pooled output

        def down_bwd(rec, dskip, dpooled, k, add=None):
            self._temb_grad(dskip, dtemb, k, dev)
            h, w = rec["hw"]
            cout = rec["cout"]
            dx = V(torch.empty((B, h, w, cout), device=dev, dtype=torch.float32))
            call("ddpm_pool_skip_bwd", dskip.ptr, dskip.ld, dpooled.ptr, dpooled.ld, _ptr(rec["argmax"]), dx.ptr, dx.ld, B, h, w, cout,
                 _stream())
            if not v1:
                dx = self._attention_bwd(V, rec["attn"], dx, grads)
            return self._double_conv_bwd(V, rec["dc"], dx, grads, add=add)

        dp2 = down_bwd(tape["enc3"], dskips["dec1"], dp3, 2)
        dskip1 = down_bwd(tape["enc2"], dskips["dec2"], dp2, 1, add=dskips["dec3"])   # total gradient of skip1
        self._temb_grad(dskip1, dtemb, 0, dev)
        self._double_conv_bwd(V, tape["enc1"], dskip1, grads, need_dsrc=False)

        # time / label embedding Linears
        y = tape["y"]
        chans = [bc, 2 * bc, 4 * bc, 8 * bc]
        g = EmbedBwdArgs()
        g.batch, g.emb_dim, g.n_labels = B, self.emb, (0 if y is None else y.numel())
        g.emb = tape["emb"].data_ptr()
        outs = {}
        if y is not None:
            g.y = y.data_ptr()
            g.label_table = self.T("label_embedding.weight").data_ptr()
            g.n_classes = self.T("label_embedding.weight").shape[0]
            outs["label_embedding.weight"] = self._gnew(self.T("label_embedding.weight").shape, dev, zero=True)
            g.dlabel_table = outs["label_embedding.weight"].data_ptr()
        for k in range(4):
            g.ch[k] = chans[k]
            g.dtemb[k] = dtemb[k].data_ptr()
            for nm, field in ((f"time_fc{k + 1}.weight", g.dw), (f"time_fc{k + 1}.bias", g.db)):
                outs[nm] = self._gnew(self.T(nm).shape, dev)
                field[k] = outs[nm].data_ptr()
            if y is not None:
                g.lw[k] = self.T(f"label_emdedding_fc{k + 1}.weight").data_ptr()
                for nm, field in ((f"label_emdedding_fc{k + 1}.weight", g.dlw), (f"label_emdedding_fc{k + 1}.bias", g.dlb)):
                    outs[nm] = self._gnew(self.T(nm).shape, dev)
                    field[k] = outs[nm].data_ptr()
        call("ddpm_time_embed_bwd", C.byref(g), _stream())
        grads.update(outs)
        self._gflush()
        if self._s2 is not None:
            torch.cuda.current_stream().wait_stream(self._s2)
        if scaled:
            own = [t for t in self._gall if t.numel()]
            if own:
                n = len(own)
                call("ddpm_scale_tensors", n, (C.c_void_p * n)(*[t.data_ptr() for t in own]), (C.c_int64 * n)(*[t.numel() for t in own]),
                     C.c_void_p(st.data_ptr() + 8), _stream())
            if self.grad_sink is not None:
                self.grad_sink.post_scale = st[2:3].clone()     # the averager's slices: unscaled when it finishes
        self._gall = []
        return grads

    def _pack_convT_dgrad(self, wt):
        out = torch.empty(packed_floats(4, wt.shape[0], wt.shape[1]), device=wt.device, dtype=torch.float32)
        call("ddpm_pack_convT2x2_dgrad", _ptr(wt.detach().contiguous()), wt.shape[0], wt.shape[1], _ptr(out), _stream())
        return out


class UNetFunction(torch.autograd.Function):
    """The whole UNet as one autograd node. Inputs after `engine` are (x, t, y, *parameters)."""

    @staticmethod
    def forward(ctx, engine, x, t, y, names, *params):
        out, fms, tape = engine.forward_train(x, t, y)
        ctx.engine, ctx.tape, ctx.names = engine, tape, names
        ctx.mark_non_differentiable(*fms)
        ctx.set_materialize_grads(False)   # no zero-filled gradient tensors for the feature maps
        return (out, *fms)

    @staticmethod
    def backward(ctx, dout, *unused):
        if dout is None:                   # nothing downstream depends on the prediction
            return (None,) * (5 + len(ctx.names))
        eng = ctx.engine
        with eng.math_scope():
            grads = eng.backward(ctx.tape, dout)
        if eng.range_watching():
            eng.range_post()       # examined at the start of the next call (no host wait here)
        ctx.tape = None
        sink = eng.grad_sink
        if sink is not None and sink.takes_over(grads):
            # the averager owns the gradients (slices of its flat buffer, buckets already in flight): it assigns
            # .grad itself when it finishes, so autograd gets nothing to accumulate (and nothing to copy)
            return (None,) * (5 + len(ctx.names))
        return (None, None, None, None, None, *[grads.get(n) for n in ctx.names])
