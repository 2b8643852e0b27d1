"""Executor: compiles a flattened layer list into fused stages and runs forward / backward / the
gradient-penalty linearised forward on the HIP kernels.  There is no tape: every backward formula is
written out (SURVEY.md 8a, rows T1-T8 and the GP second-order derivation).

A stage = one linear op (Dense | Conv2D | Conv2DTranspose) [+ bias] [+ BatchNormalization] [+ LeakyReLU |
tanh] [+ Dropout]; Reshape / Flatten are views.  Stage fusion on the device:
  conv (+bias) + LeakyReLU + Dropout      -> one launch (epilogue of the MFMA kernel)
  dgrad + LeakyReLU'/Dropout' of the stage below -> one launch (BG_EPI_MUL_GRAD epilogue)
  BatchNormalization + LeakyReLU          -> stats pass + one apply pass
"""
from __future__ import annotations

import os
from typing import List, Optional

import numpy as np
import torch

from . import dist, ops
from ._lib import EPI_NONE, EPI_BIAS_LRELU, EPI_MUL_GRAD, EPI_TANH, EPI_AFFINE_LRELU


class Stage:
    def __init__(self, lin, in_shape):
        self.lin = lin
        self.kind = lin.kind
        self.in_shape = tuple(in_shape)
        self.out_shape = tuple(lin.out_shape(in_shape))
        self.bn = None
        self.act = getattr(lin, "activation", None)
        self.alpha = 1.0
        self.drop = None

    @property
    def fusable_grad(self):
        return self.bn is None and self.act == "lrelu"


class Context:
    """Activation / gradient buffers of one pass at a fixed batch size (caller-owned HBM)."""

    def __init__(self, net, B, device, drop_rows=None):
        self.B = B
        # Dropout applies to the first ``drop_rows`` samples of the batch only (default: all).  The critic step runs
        # [fakes; reals] (training=True) and x-hat (training=False) through ONE pass: 2B rows with masks, B without.
        self.drop_rows = B if drop_rows is None else int(drop_rows)
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=device)
        self.a0 = f(B, *net.in_shape)
        self.din = None
        self.a, self.z, self.keep, self.mean, self.inv, self.dz, self.v = [], [], [], [], [], [], []
        self.xin: List[Optional[torch.Tensor]] = [None] * len(net.stages)
        for st in net.stages:
            self.a.append(f(B, *st.out_shape))
            self.dz.append(None)
            self.v.append(None)
            self.z.append(f(B, *st.out_shape) if st.bn is not None else None)
            c = st.out_shape[-1]
            self.mean.append(f(c) if st.bn is not None else None)
            self.inv.append(f(c) if st.bn is not None else None)
            self.keep.append(None)
        # dropout masks of all stages live in ONE uint8 buffer (16-byte aligned slices): a pass draws them with one launch
        DR = self.drop_rows
        sizes = [int(np.prod((DR,) + st.out_shape)) if st.drop else 0 for st in net.stages]
        offs, total = [], 0
        for n in sizes:
            offs.append(total)
            total += -(-n // 16) * 16
        self.keep_flat = torch.empty(max(total, 16), dtype=torch.uint8, device=device)
        self.keep_total = total
        for i, (st, n) in enumerate(zip(net.stages, sizes)):
            if n:
                self.keep[i] = self.keep_flat[offs[i]:offs[i] + n].view((DR,) + st.out_shape)
        rates = {float(st.drop) for st in net.stages if st.drop}
        self.keep_rate = rates.pop() if len(rates) == 1 else None      # None: stages differ, draw per stage
        self._net, self._device = net, device
        self._extra = {}
        self.dropout_active = False

    def keep_elems(self, i):
        """Output elements of stage i covered by its dropout mask (0 = all of them): the bg_epilogue.keep_elems field."""
        return 0 if self.drop_rows >= self.B else self.drop_rows * int(np.prod(self._net.stages[i].out_shape))

    def rows(self, lo, hi):
        """A view of this pass restricted to samples [lo, hi): what gp_second_order needs of the x-hat rows of a merged pass."""
        return _RowView(self, lo, hi)

    def buf(self, lst, i):
        if lst[i] is None:
            lst[i] = torch.empty((self.B,) + self._net.stages[i].out_shape, dtype=torch.float32, device=self._device)
        return lst[i]

    def bn_partials(self, i, rows, C):
        """Room for the per-workgroup BatchNorm statistics the conv of stage i leaves behind (bg_epilogue.stats)."""
        key = ("bn_partials", i)
        n = rows * 2 * C
        if key not in self._extra or self._extra[key].numel() < n:
            self._extra[key] = torch.empty(n, dtype=torch.float32, device=self._device)
        return self._extra[key]

    def bn_sums(self, i, C):
        key = ("bn_sums", i)
        if key not in self._extra:
            self._extra[key] = torch.empty(2 * C, dtype=torch.float32, device=self._device)
        return self._extra[key]

    def input_grad(self):
        if self.din is None:
            self.din = torch.empty((self.B,) + self._net.in_shape, dtype=torch.float32, device=self._device)
        return self.din


class _RowView:
    """Samples [lo, hi) of a Context: sliced activations / gradients, own scratch for the linearised forward."""

    def __init__(self, ctx, lo, hi):
        self._ctx, self.lo, self.hi, self.B = ctx, lo, hi, hi - lo
        self.a = [t[lo:hi] for t in ctx.a]
        self.dz = [None if t is None else t[lo:hi] for t in ctx.dz]
        key = ("rowview_v", lo, hi)
        self.v = ctx._extra.setdefault(key, [None] * len(ctx.a))
        self._net, self._device = ctx._net, ctx._device

    def buf(self, lst, i):
        if lst[i] is None:
            lst[i] = torch.empty((self.B,) + self._net.stages[i].out_shape, dtype=torch.float32, device=self._device)
        return lst[i]


class Net:
    def __init__(self, flat_layers, store):
        self.store = store
        self.device = store.device
        self.blur = None
        self.capture_branches = None     # test instrumentation (gp_second_order_merged): a list collects the x-hat rows' LeakyReLU signs
        self.stages: List[Stage] = []
        self.in_shape = tuple(flat_layers[0][1])
        cur = None
        for idx, (l, s) in enumerate(flat_layers):
            k = l.kind
            if k == "blur":
                if idx != 0:
                    raise NotImplementedError("GaussianBlur2D is only supported as the first layer (blurred_gan.py:31-34)")
                self.blur = l
            elif k in ("dense", "conv", "convT"):
                cur = Stage(l, s)
                self.stages.append(cur)
            elif k == "bn":
                if cur is None or cur.bn is not None or cur.act is not None or cur.drop:
                    raise NotImplementedError("BatchNormalization must directly follow a linear layer")
                cur.bn = l
            elif k == "lrelu":
                if cur is None or cur.act is not None or cur.drop:
                    raise NotImplementedError("LeakyReLU must follow a linear layer or its BatchNormalization")
                cur.act, cur.alpha = "lrelu", l.alpha
            elif k == "dropout":
                if cur is None or cur.drop or cur.bn is not None or cur.act != "lrelu":
                    raise NotImplementedError("Dropout is supported after conv + LeakyReLU (demo_celeba.py:99-121)")
                cur.drop = l.rate
            elif k in ("reshape", "flatten"):
                pass
            else:
                raise NotImplementedError(f"layer kind {k!r}")
        for st in self.stages:
            if st.kind == "dense" and st.bn is None and st.act is not None:
                raise NotImplementedError("Dense + activation without BatchNormalization is not on the reference path")
            if st.act == "tanh" and st.bn is not None:
                raise NotImplementedError("tanh after BatchNormalization is not on the reference path")
        self.out_shape = self.stages[-1].out_shape
        self.conv_layers = [st.lin for st in self.stages if st.kind in ("conv", "convT")]
        self._ctx = {}
        self._ws = None
        self._taps_cache = {}
        self._splitk_bytes = {}
        self.rng_offset = 0
        self.sync_bn = True       # data parallel: BatchNormalization statistics over the global batch
        # statistics in the producing conv's epilogue: opt-in since round 3 -- the plain gather-GEMM variant now stores float4
        # (transposed accumulators), the statistics variant cannot, and the separate statistics pass costs less than that (C2
        # +0.24 %, C4 +0.3 % in a same-box A/B); BGAN_FUSED_BN_STATS=1 selects the fused form
        self.bn_bwd_read_y = os.environ.get("BGAN_BN_BWD_READ_Y") == "1"
        self.fuse_bn_stats = os.environ.get("BGAN_FUSED_BN_STATS") == "1" and not os.environ.get("BGAN_NO_FUSED_BN_STATS")

    # ------------------------------------------------------------------ resources
    def context(self, B, tag="default", drop_rows=None) -> Context:
        key = (B, tag, drop_rows)
        if key not in self._ctx:
            self._ctx[key] = Context(self, B, self.device, drop_rows=drop_rows)
        return self._ctx[key]

    def workspace(self, nbytes):
        n = max(int(nbytes), 1024)
        if self._ws is None or self._ws.numel() * 4 < n:
            self._ws = torch.empty((n + 3) // 4 + 1024, dtype=torch.float32, device=self.device)
        return self._ws

    def _epi(self, bwd_data, B, H, W, Cin, Cout, k, stride, mode, **kw):
        """Epilogue descriptor for one conv launch; attaches split-K scratch when the library's plan wants it."""
        key = (bool(bwd_data), B, H, W, Cin, Cout, k, stride)
        nb = self._splitk_bytes.get(key)
        if nb is None:
            nb = self._splitk_bytes[key] = ops.conv2d_splitk_workspace_bytes(bwd_data, B, H, W, Cin, Cout, k, stride)
        return ops.epilogue(mode, ws=self.workspace(nb) if nb else None, **kw)

    def prepare_weights(self):
        """(Re)builds the transposed kernel copies the 'other direction' of each conv needs."""
        self.store.refresh_transposed(self.conv_layers)

    def blur_taps(self, H, W):
        """(device buffer, tap count) of the layer's CURRENT std for H x W images (gaussian_blur.py:21-31,83-88 through
        bg_blur_policy / bg_gauss_kernel_1d).  The taps of one image size live in ONE persistent device buffer whose address never
        changes (a recorded step program holds it); when std has changed since the last call the new weights are copied into it
        in stream order.  sigma changes every batch under BlurDecayController: the upload goes through a pinned staging ring and
        an asynchronous copy (a pageable copy would wait for the stream to drain)."""
        std = float(self.blur.std)
        ent = self._taps_cache.get((H, W))
        if ent is None:
            ent = self._taps_cache[(H, W)] = {"std": None, "nt": 0, "dev": torch.zeros(1024, dtype=torch.float32, device=self.device)}
        if ent["std"] != std:
            ks, se, nt = ops.blur_policy(std, H, W)
            taps = ops.gauss_kernel_1d(se, ks)
            assert len(taps) == nt
            if nt > 1024:
                raise NotImplementedError(f"blur with {nt} taps (> 1024)")
            ring = self.__dict__.setdefault("_taps_ring", {"i": 0, "bufs": [], "evs": []})
            if not ring["bufs"]:
                for _ in range(8):
                    ring["bufs"].append(torch.empty(1024, dtype=torch.float32, pin_memory=self.device.type == "cuda"))
                    ring["evs"].append(None)
            i = ring["i"] = (ring["i"] + 1) % len(ring["bufs"])
            if ring["evs"][i] is not None:
                ring["evs"][i].synchronize()                  # slot re-used 8 uploads later: normally long complete
            host = ring["bufs"][i][:nt]
            host.copy_(torch.tensor(taps, dtype=torch.float32))
            ent["dev"][:nt].copy_(host, non_blocking=True)
            if self.device.type == "cuda":
                ev = torch.cuda.Event()
                ev.record()
                ring["evs"][i] = ev
            ent["std"], ent["nt"] = std, nt
        return ent["dev"], ent["nt"]

    def blur_n_taps(self, H=None, W=None):
        """Tap count of the current std (uploads the taps if std changed): part of a step program's key."""
        if self.blur is None:
            return 0
        H = self.in_shape[0] if H is None else H
        W = self.in_shape[1] if W is None else W
        return self.blur_taps(H, W)[1]

    def apply_blur(self, x, y):
        """y = blur(x) with the layer's current std (blurred_gan.py:30; self-adjoint, so also its backward)."""
        B, H, W, C = x.shape
        taps, nt = self.blur_taps(H, W)
        nb = ops.blur_workspace_bytes(B, H, W, C, nt)
        tmp = self.workspace(nb) if nb else None
        with ops.trace_range("blur"):
            return ops.blur_nhwc(x, y, taps, nt, tmp)

    # ------------------------------------------------------------------ forward
    def forward(self, ctx: Context, inputs, training=False, masks=None, seed=0, lerp_alpha=None, lerp_out=None):
        """inputs: a tensor [B,...] or a list of tensors concatenated along the batch.  ``training`` switches
        Dropout and BatchNormalization exactly as Keras' ``training=`` argument does.
        ``lerp_alpha`` (with inputs = [f, r]): a third batch slice x-hat = r + alpha (f - r) follows the two (wgan.py:239).  Behind
        a blur layer of a geometry the row-block kernel takes, all three slices are blurred by ONE launch that forms x-hat on the
        fly (bg_blur3_lerp_nhwc_f32); otherwise x-hat is written to ``lerp_out`` by bg_lerp_f32 and joins the list."""
        if self.store.tr_dirty:
            self.prepare_weights()
        if not isinstance(inputs, (list, tuple)):
            inputs = [inputs]
        B = ctx.B
        fused3 = False
        if lerp_alpha is not None:
            f, r = inputs
            n = int(f.shape[0])
            assert 3 * n == B
            if self.blur is not None:
                H, W, C = self.in_shape
                taps, nt = self.blur_taps(H, W)
                fused3 = not os.environ.get("BGAN_NO_FUSED_BLUR3") and ops.blur3_lerp_supported(n, H, W, C, nt)
            if fused3:
                with ops.trace_range("blur"):
                    ops.blur3_lerp(f.view(n, *self.in_shape), r.view(n, *self.in_shape), lerp_alpha, ctx.a0, taps, nt)
            else:
                inputs = [f, r, ops.lerp(r, f, lerp_alpha, lerp_out)]
        assert fused3 or sum(int(t.shape[0]) for t in inputs) == B, "batch mismatch"
        if fused3:
            x = ctx.a0
        elif self.blur is not None:
            o = 0
            for t in inputs:
                n = int(t.shape[0])
                self.apply_blur(t.view(n, *self.in_shape), ctx.a0[o:o + n])
                o += n
            x = ctx.a0
        elif len(inputs) == 1:
            x = inputs[0]
        else:
            o = 0
            for t in inputs:
                n = int(t.shape[0])
                ops.copy_(ctx.a0[o:o + n], t.view(n, *self.in_shape).contiguous())
                o += n
            x = ctx.a0
        ctx.dropout_active = bool(training)
        mi = 0
        drawn = False
        if training and masks is None and ctx.keep_rate is not None and ctx.keep_total:
            ops.keep_mask(ctx.keep_flat[:ctx.keep_total], 1.0 - ctx.keep_rate, seed, counter=(self, "rng_offset"))
            drawn = True
        folded = ()
        if not training:
            # inference BatchNorms (the generator inside the D-step): every layer's fold in ONE launch ahead of the first conv
            # instead of a tiny launch between each pair of convs (two kernel boundaries where one would do)
            fl = [(j, s2) for j, s2 in enumerate(self.stages)
                  if s2.bn is not None and s2.kind != "dense" and s2.lin.vars.get("bias") is None]
            if 1 < len(fl) <= ops.FOLD_MAX and not os.environ.get("BGAN_NO_FOLD_MANY"):
                args = []
                for j, s2 in fl:
                    Cj = s2.out_shape[-1]
                    sc = ctx.bn_sums(j, Cj)
                    v = s2.bn.vars
                    args.append((v["gamma"], v["beta"], v["moving_mean"], v["moving_variance"], s2.bn.epsilon, sc[:Cj], sc[Cj:]))
                ops.bn_fold_many(args)
                folded = {j for j, _ in fl}
        for i, st in enumerate(self.stages):
            xin = x.view(B, *st.in_shape)
            ctx.xin[i] = xin
            out = ctx.a[i]
            tgt = ctx.z[i] if st.bn is not None else out
            bias = st.lin.vars.get("bias")
            keep = None
            if st.drop:
                if training:
                    keep = ctx.keep[i]
                    if masks is not None:
                        keep.copy_(masks[mi].view(keep.shape))
                    elif not drawn:
                        ops.keep_mask(keep, 1.0 - st.drop, seed, counter=(self, "rng_offset"))
                mi += 1
            stat_rows = 0          # rows of BatchNorm statistics partials the conv left behind (0: none)
            if st.kind == "dense":
                K, N = st.in_shape[0], st.out_shape[0]
                ops.gemm(xin, st.lin.vars["kernel"], tgt, B, N, K, bias=bias)
            else:
                lin = st.lin
                if st.kind == "conv":
                    geom = (False, B, st.in_shape[0], st.in_shape[1], st.in_shape[2], lin.filters, lin.k, lin.stride)
                else:           # ConvT forward == data-gradient of the conv whose input side is this layer's output
                    geom = (True, B, st.out_shape[0], st.out_shape[1], lin.filters, st.in_shape[2], lin.k, lin.stride)
                fold_bn = st.bn is not None and not training and bias is None
                if fold_bn:
                    # inference BatchNorm (generator inside the D-step) folded into the conv epilogue: no z, no BN pass
                    C = st.out_shape[-1]
                    sc = ctx.bn_sums(i, C)
                    bn = st.bn
                    if i not in folded:
                        ops.bn_fold(bn.vars["gamma"], bn.vars["beta"], bn.vars["moving_mean"], bn.vars["moving_variance"], bn.epsilon,
                                    sc[:C], sc[C:])
                    epi = self._epi(*geom, EPI_AFFINE_LRELU, bias=sc[C:], ref=sc[:C], alpha=st.alpha)
                    tgt = out
                elif st.bn is not None:
                    # training BatchNorm behind a bias-free conv: the MFMA kernel leaves the column sums of what it stores
                    # (one row per workgroup); at most one workgroup per 32 output rows and sub-pixel phase
                    stats = None
                    # under SyncBN the statistics ride in the conv's epilogue (the separate statistics pass would be two more
                    # launches on the forward critical path beside every exchange); single-device: opt-in, see __init__
                    if training and bias is None and (self.fuse_bn_stats or (self.sync_bn and dist.collectives_active()
                                                                           and not os.environ.get("BGAN_NO_FUSED_BN_STATS"))):
                        C = st.out_shape[-1]
                        stats = ctx.bn_partials(i, int(np.prod((B,) + st.out_shape[:-1])) // 32 + 64, C)
                    epi = self._epi(*geom, EPI_NONE, bias=bias, stats=stats)
                elif st.act == "lrelu":
                    epi = self._epi(*geom, EPI_BIAS_LRELU, bias=bias, keep=keep, alpha=st.alpha,
                                    scale=1.0 / (1.0 - st.drop) if st.drop else 1.0,
                                    keep_elems=ctx.keep_elems(i) if keep is not None else 0)
                elif st.act == "tanh":
                    epi = self._epi(*geom, EPI_TANH, bias=bias)
                else:
                    epi = self._epi(*geom, EPI_NONE, bias=bias)
                if st.kind == "conv":
                    ops.conv2d_fwd(xin, self.store.transposed_kernel(st.lin), tgt, st.lin.k, st.lin.stride, epi)
                else:   # Conv2DTranspose forward == data-gradient of the conv with the same kernel array
                    ops.conv2d_bwd_data(xin, st.lin.vars["kernel"], tgt, st.lin.k, st.lin.stride, epi)
                stat_rows = ops.conv2d_stats_rows(epi) if st.bn is not None else 0
            if st.bn is not None and not (st.kind != "dense" and not training and bias is None):
                C = st.out_shape[-1]
                M = tgt.numel() // C
                bn = st.bn
                if training and self.sync_bn and dist.collectives_active():
                    # SyncBN: batch statistics over the GLOBAL batch -- all-reduce the per-channel sums
                    sums = ctx.bn_sums(i, C)
                    if stat_rows:
                        ops.bn_sums_from_partials(ctx.bn_partials(i, stat_rows, C), stat_rows, C, sums)
                    else:
                        ws = self.workspace(ops._lib.load().bg_bn_workspace_bytes(M, C))
                        ops.bn_stats(tgt, M, C, sums, ws)
                    dist.all_reduce_sum_(sums)
                    # finalize + apply in one launch: with the statistics out of the conv's epilogue the chain around the
                    # exchange is two launches per layer (partials -> sums, this)
                    ops.bn_finalize_apply(sums, M * dist.world_size(), tgt, out, M, C, bn.vars["gamma"], bn.vars["beta"], ctx.mean[i],
                                          ctx.inv[i], bn.vars["moving_mean"], bn.vars["moving_variance"], eps=bn.epsilon,
                                          momentum=bn.momentum, unbiased=(len(st.out_shape) == 3), lrelu_alpha=st.alpha)
                elif training and stat_rows:
                    ops.bn_train_fwd_partials(ctx.bn_partials(i, stat_rows, C), stat_rows, tgt, out, M, C, bn.vars["gamma"], bn.vars["beta"],
                                              bn.vars["moving_mean"], bn.vars["moving_variance"], ctx.mean[i], ctx.inv[i], eps=bn.epsilon,
                                              momentum=bn.momentum, unbiased=(len(st.out_shape) == 3), lrelu_alpha=st.alpha)
                elif training:
                    ws = self.workspace(ops._lib.load().bg_bn_workspace_bytes(M, C))
                    ops.bn_train_fwd(tgt, out, M, C, bn.vars["gamma"], bn.vars["beta"], bn.vars["moving_mean"],
                                     bn.vars["moving_variance"], ctx.mean[i], ctx.inv[i], ws, eps=bn.epsilon,
                                     momentum=bn.momentum, unbiased=(len(st.out_shape) == 3), lrelu_alpha=st.alpha)
                else:
                    ops.bn_infer_fwd(tgt, out, M, C, bn.vars["gamma"], bn.vars["beta"], bn.vars["moving_mean"],
                                     bn.vars["moving_variance"], eps=bn.epsilon, lrelu_alpha=st.alpha)
            x = out
        return x

    def predict(self, x, training=False):
        B = int(x.shape[0])
        ctx = self.context(B, "predict")
        x = x.to(self.device, torch.float32).contiguous()
        return self.forward(ctx, x, training=training, seed=np.random.randint(1 << 30)).clone()

    def _bn_y(self, ctx, i):
        """The saved activation the BatchNorm backward passes read for the LeakyReLU sign -- None by default: the kernels re-derive
        the sign from z with the forward's own expression (bit-identical, one tensor less per pass).  BGAN_BN_BWD_READ_Y=1 reads it."""
        return ctx.a[i] if self.bn_bwd_read_y else None

    # ------------------------------------------------------------------ backward
    def backward(self, ctx: Context, dout, need_dx=False, need_dw=True, beta=0.0, scale=1.0, reducer=None, dw_rows=None,
                 dx_rows=None, defer_conv_dw=False):
        """Reverse pass of the last ``forward`` on ``ctx``.  Weight gradients go to ``store.grad``
        (= beta*old + scale*new).  Returns d(loss)/d(net input) *before* the blur (or None).
        ``reducer`` (dist.GradReducer): this pass completes the gradients, so each stage's slice of the flat buffer is
        handed to the bucketed all-reduce as soon as its kernels are enqueued.
        ``dw_rows``: weight gradients from the first dw_rows samples only; ``dx_rows`` = (lo, hi): the input gradient (last
        data-gradient + blur^T) for those samples only -- the merged critic pass wants weights from [fakes; reals] and the
        image gradient of x-hat.  ``defer_conv_dw``: the conv filter gradients are left to ``gp_second_order_merged`` (bias
        and Dense gradients are still taken here)."""
        st_ = self.store
        if need_dw:
            st_.ensure_opt_state()
        B = ctx.B
        WB = B if dw_rows is None else int(dw_rows)
        g, g_is_dz = dout, False
        sync_bn = self.sync_bn and dist.collectives_active()
        pending_stats = None        # SyncBN: the all-reduce of the NEXT (lower) stage's backward statistics, in flight
        deferred_ready = []         # SyncBN overlap: gradient slices finished but not yet handed to the reducer (see weight_grads)
        for i in range(len(self.stages) - 1, -1, -1):
            st = self.stages[i]
            gv = g.view(B, *st.out_shape)
            # ---- through activation / BN -> gradient w.r.t. the linear op's output
            if st.bn is not None:
                C = st.out_shape[-1]
                M = gv.numel() // C
                dz = ctx.buf(ctx.dz, i)
                ws = self.workspace(ops._lib.load().bg_bn_workspace_bytes(M, C))
                dg = st_.grad_of(st.bn, "gamma") if need_dw else ctx.bn_sums(("dgamma_scratch", i), C)[:C]
                db = st_.grad_of(st.bn, "beta") if need_dw else ctx.bn_sums(("dbeta_scratch", i), C)[:C]
                if sync_bn:
                    world = dist.world_size()
                    sums = ctx.bn_sums(i, C)
                    if pending_stats is None:       # last stage: nothing above it to overlap with
                        ops.bn_bwd_stats(gv, self._bn_y(ctx, i), ctx.z[i], M, C, ctx.mean[i], ctx.inv[i], sums, ws, lrelu_alpha=st.alpha,
                                         gamma=st.bn.vars["gamma"], beta=st.bn.vars["beta"])
                        pending_stats = dist.all_reduce_sum_async(sums)
                    pending_stats.wait()            # issued before the filter gradient of the stage above: it travelled meanwhile
                    pending_stats = None
                    ops.bn_bwd_apply(gv, self._bn_y(ctx, i), ctx.z[i], dz, M, M * world, C, st.bn.vars["gamma"], ctx.mean[i], ctx.inv[i], sums,
                                     lrelu_alpha=st.alpha, beta=st.bn.vars["beta"])
                    if need_dw:     # the sums are already global: pre-divide so the flat gradient SUM all-reduce restores them
                        ops.bn_param_grads(sums, C, 1.0 / world, dg, db)
                else:
                    ops.bn_train_bwd(gv, self._bn_y(ctx, i), ctx.z[i], dz, M, C, st.bn.vars["gamma"], ctx.mean[i], ctx.inv[i], dg, db, ws,
                                     lrelu_alpha=st.alpha, beta=st.bn.vars["beta"])
            elif st.act == "lrelu":
                if g_is_dz:
                    dz = gv
                else:
                    keep = ctx.keep[i] if (st.drop and ctx.dropout_active) else None
                    dzb = ctx.buf(ctx.dz, i)
                    if keep is not None and ctx.drop_rows < B:      # masked rows first, the unmasked tail separately
                        DR = ctx.drop_rows
                        ops.mul_grad(gv[:DR], ctx.a[i][:DR], dzb[:DR], keep=keep, alpha=st.alpha, scale=1.0 / (1.0 - st.drop))
                        ops.mul_grad(gv[DR:], ctx.a[i][DR:], dzb[DR:], keep=None, alpha=st.alpha, scale=1.0)
                        dz = dzb
                    else:
                        dz = ops.mul_grad(gv, ctx.a[i], dzb, keep=keep, alpha=st.alpha,
                                          scale=1.0 / (1.0 - st.drop) if (st.drop and ctx.dropout_active) else 1.0)
            elif st.act == "tanh":
                dz = ops.tanh_bwd(gv, ctx.a[i], ctx.buf(ctx.dz, i))
            else:
                dz = gv
            ctx.dz[i] = dz
            xin = ctx.xin[i]
            lin = st.lin
            prev = self.stages[i - 1] if i > 0 else None

            def weight_grads(defer_ready=False):
                dW = st_.grad_of(lin, "kernel")
                xw, dzw = (xin, dz) if WB == B else (xin[:WB], dz.view(B, *st.out_shape)[:WB])
                if st.kind == "dense":
                    K, N = st.in_shape[0], st.out_shape[0]
                    ops.gemm(xw, dzw, dW, K, N, WB, transA=True, beta=beta, scale=scale)
                    rows = WB
                elif st.kind == "conv":
                    if not defer_conv_dw:
                        Bc, H, W, Ci = xw.shape
                        nb = ops.conv2d_bwd_filter_workspace_bytes(Bc, H, W, Ci, lin.filters, lin.k, lin.stride)
                        ops.conv2d_bwd_filter(xw, dzw, dW, lin.k, lin.stride, beta, scale, self.workspace(nb) if nb else None)
                    rows = dzw.numel() // lin.filters
                else:
                    Bc, H, W, Ci = dzw.shape          # conv input side == ConvT output
                    nb = ops.conv2d_bwd_filter_workspace_bytes(Bc, H, W, Ci, xw.shape[3], lin.k, lin.stride)
                    ops.conv2d_bwd_filter(dzw, xw, dW, lin.k, lin.stride, beta, scale, self.workspace(nb) if nb else None)
                    rows = dzw.numel() // lin.filters
                if "bias" in lin.vars:
                    N = st.out_shape[-1]
                    ws = self.workspace(ops.colsum_workspace_bytes(rows, N))
                    ops.colsum(dzw, st_.grad_of(lin, "bias"), rows, N, ws, beta=beta, scale=scale)
                if reducer is not None and not (defer_conv_dw and st.kind == "conv"):
                    # Under the SyncBN overlap a slice is NOT handed over right away: ready() may flush a 16 MB bucket onto the
                    # collective stream, and the small statistics exchange of the next stage down, issued one iteration later,
                    # would queue behind it -- back on the critical path.  The slice waits until that exchange is in flight.
                    if defer_ready:
                        deferred_ready.append(st_.train_range(lin, st.bn))
                    else:
                        reducer.ready(*st_.train_range(lin, st.bn))

            # SyncBN with a BatchNorm stage below: the data gradient goes FIRST, the lower stage's backward statistics are
            # reduced and their all-reduce is put in flight, and only then this stage's filter gradient runs -- the small
            # latency-bound collective hides behind an MFMA kernel instead of sitting on the critical path of every layer.
            # (The filter gradient's split-K workspace and the statistics' partials share net.workspace(): the statistics
            # kernel has finished with it before the filter gradient is enqueued on the same stream.)
            overlap = sync_bn and need_dw and prev is not None and prev.bn is not None
            if need_dw and not overlap:
                weight_grads()
            # ---- input gradient
            if i == 0 and not need_dx:
                if reducer is not None:
                    for rng in deferred_ready:
                        reducer.ready(*rng)
                return None
            fuse = prev is not None and prev.fusable_grad and st.kind != "dense"
            tgt = ctx.buf(ctx.dz, i - 1).view(B, *st.in_shape) if i > 0 else ctx.input_grad().view(B, *st.in_shape)
            Bx = B
            dzx = dz
            if i == 0 and dx_rows is not None:        # the image gradient is wanted for samples [lo, hi) only
                lo, hi = dx_rows
                dzx = dz.view(B, *st.out_shape)[lo:hi]
                tgt = tgt[lo:hi]
                Bx = hi - lo
            epi = None
            if st.kind != "dense":
                if st.kind == "conv":
                    geom = (True, Bx, st.in_shape[0], st.in_shape[1], st.in_shape[2], lin.filters, lin.k, lin.stride)
                else:
                    geom = (False, Bx, st.out_shape[0], st.out_shape[1], lin.filters, st.in_shape[2], lin.k, lin.stride)
                if fuse:
                    pk = ctx.keep[i - 1] if (prev.drop and ctx.dropout_active) else None
                    epi = self._epi(*geom, EPI_MUL_GRAD, ref=ctx.a[i - 1], keep=pk, alpha=prev.alpha,
                                    scale=1.0 / (1.0 - prev.drop) if pk is not None else 1.0,
                                    keep_elems=ctx.keep_elems(i - 1) if pk is not None else 0)
                else:
                    epi = self._epi(*geom, EPI_NONE)
            if st.kind == "dense":
                K, N = st.in_shape[0], st.out_shape[0]
                ops.gemm(dzx, lin.vars["kernel"], tgt, Bx, K, N, transB=True)
            elif st.kind == "conv":
                ops.conv2d_bwd_data(dzx, lin.vars["kernel"], tgt, lin.k, lin.stride, epi)
            else:
                ops.conv2d_fwd(dzx, self.store.transposed_kernel(lin), tgt, lin.k, lin.stride, epi)
            if overlap:
                Cp = prev.out_shape[-1]
                gp = tgt.view(B, *prev.out_shape)
                Mp = gp.numel() // Cp
                sums = ctx.bn_sums(i - 1, Cp)
                ops.bn_bwd_stats(gp, self._bn_y(ctx, i - 1), ctx.z[i - 1], Mp, Cp, ctx.mean[i - 1], ctx.inv[i - 1], sums,
                                 self.workspace(ops._lib.load().bg_bn_workspace_bytes(Mp, Cp)), lrelu_alpha=prev.alpha,
                                 gamma=prev.bn.vars["gamma"], beta=prev.bn.vars["beta"])
                pending_stats = dist.all_reduce_sum_async(sums)
                if reducer is not None:             # slices finished before this exchange was issued may go out behind it now
                    for rng in deferred_ready:
                        reducer.ready(*rng)
                    deferred_ready.clear()
                weight_grads(defer_ready=True)
            g, g_is_dz = tgt, fuse
        if reducer is not None:
            for rng in deferred_ready:
                reducer.ready(*rng)
        din = g
        if self.blur is not None:
            # forward's blurred input is dead by now; reuse it for blur^T(din)
            out = ctx.a0 if dx_rows is None else ctx.a0[dx_rows[0]:dx_rows[1]]
            return self.apply_blur(din.view(din.shape[0], *self.in_shape), out)
        return din

    # ------------------------------------------------------------------ GP second order
    def gp_second_order_merged(self, ctx: Context, lo, hi, reducer=None):
        """The second order of the penalty AND the filter gradients of the whole merged critic pass, one launch per layer.

        The first-order gradient of layer i is wgrad(x = a_{i-1}[rows of fakes, reals], dy = dz_i[same rows]); the penalty adds
        wgrad(x = delta-bar_{i-1}, dy = zeta_i) over the x-hat rows (GP derivation step 2).  Rows are the
        contraction index, so with delta-bar_{i-1} written INTO the x-hat rows [lo, hi) of the pass's activation buffer a_{i-1}
        -- dead after the backward except for their signs, which the linearised forward consumes as it overwrites them in
        place -- both are ONE filter gradient over all 3B rows of (a_{i-1}, dz_i): five launches (and their slab reduces)
        fewer per D-step, and the remaining ones run at 3B rows instead of 2B + B.  The caller has put delta-bar_0 into
        ctx.a0[lo:hi] and run ``backward(..., defer_conv_dw=True)``."""
        st_ = self.store
        st_.ensure_opt_state()
        B3, Bh = ctx.B, hi - lo
        for i, st in enumerate(self.stages):
            lin = st.lin
            last = i == len(self.stages) - 1
            if st.kind == "conv" and st.fusable_grad and not last:
                xin = ctx.xin[i]                                  # [3B, ...]: activations of [fakes; reals], delta-bar_{i-1} in the x-hat rows
                _, H, W, Ci = xin.shape
                nb = ops.conv2d_bwd_filter_workspace_bytes(B3, H, W, Ci, lin.filters, lin.k, lin.stride)
                ops.conv2d_bwd_filter(xin, ctx.dz[i].view(B3, *st.out_shape), st_.grad_of(lin, "kernel"), lin.k, lin.stride, 0.0, 1.0,
                                      self.workspace(nb) if nb else None)
                if reducer is not None:
                    reducer.ready(*st_.train_range(lin, st.bn))
                ah = ctx.a[i][lo:hi]
                if self.capture_branches is not None:            # test instrumentation: the signs about to be overwritten
                    self.capture_branches.append((ah > 0).cpu().numpy())
                epi = self._epi(False, Bh, H, W, Ci, lin.filters, lin.k, lin.stride, EPI_MUL_GRAD, ref=ah, alpha=st.alpha)
                ops.conv2d_fwd(xin[lo:hi], self.store.transposed_kernel(lin), ah, lin.k, lin.stride, epi)   # in place: ref == out
            elif st.kind == "dense" and last and st.out_shape == (1,) and st.bn is None and st.act is None:
                K = st.in_shape[0]
                ws = self.workspace(ops.colsum_workspace_bytes(Bh, K))
                ops.colsum(ctx.xin[i][lo:hi].reshape(Bh, K), st_.grad_of(lin, "kernel").view(K), Bh, K, ws, beta=1.0, scale=1.0)
                if reducer is not None:
                    reducer.ready(*st_.train_range(lin, st.bn))
            else:
                raise NotImplementedError("gradient penalty second order supports the reference critic shape only: "
                                          "[Conv2D+LeakyReLU(+Dropout)]* -> Flatten -> Dense(1) (demo_celeba.py:96-124)")

    def gp_second_order(self, ctx: Context, v0, reducer=None):
        """SURVEY.md 8a, GP derivation step 2: one linearised forward of the critic on ``v0`` with the
        LeakyReLU masks of the x-hat pass frozen, plus one wgrad per conv layer against the zeta_i kept by
        the first-order backward (``ctx.dz``); accumulates into ``store.grad`` (beta = 1)."""
        st_ = self.store
        st_.ensure_opt_state()
        B = ctx.B
        v = v0
        for i, st in enumerate(self.stages):
            lin = st.lin
            last = i == len(self.stages) - 1
            if st.kind == "conv" and st.fusable_grad and not last:
                vin = v.view(B, *st.in_shape)
                Bc, H, W, Ci = vin.shape
                nb = ops.conv2d_bwd_filter_workspace_bytes(Bc, H, W, Ci, lin.filters, lin.k, lin.stride)
                ops.conv2d_bwd_filter(vin, ctx.dz[i], st_.grad_of(lin, "kernel"), lin.k, lin.stride, 1.0, 1.0,
                                      self.workspace(nb) if nb else None)
                vo = ctx.buf(ctx.v, i)
                epi = self._epi(False, Bc, H, W, Ci, lin.filters, lin.k, lin.stride, EPI_MUL_GRAD, ref=ctx.a[i], alpha=st.alpha)
                if reducer is not None:     # this layer's gradient (first-order pass + this wgrad) is complete
                    reducer.ready(*st_.train_range(lin, st.bn))
                ops.conv2d_fwd(vin, self.store.transposed_kernel(lin), vo, lin.k, lin.stride, epi)
                v = vo
            elif st.kind == "dense" and last and st.out_shape == (1,) and st.bn is None and st.act is None:
                K = st.in_shape[0]
                ws = self.workspace(ops.colsum_workspace_bytes(B, K))
                ops.colsum(v.view(B, K), st_.grad_of(lin, "kernel").view(K), B, K, ws, beta=1.0, scale=1.0)
                if reducer is not None:
                    reducer.ready(*st_.train_range(lin, st.bn))
            else:
                raise NotImplementedError("gradient penalty second order supports the reference critic shape only: "
                                          "[Conv2D+LeakyReLU(+Dropout)]* -> Flatten -> Dense(1) (demo_celeba.py:96-124)")
