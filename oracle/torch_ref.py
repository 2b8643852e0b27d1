"""The same training step written with torch-CPU ops + autograd.

Oracle = test infrastructure (see ``oracle/__init__.py``); parity unpinned.

Two uses:
  * an independent *implementation* of the maths (library convolutions + reverse-mode AD incl.
    ``create_graph=True`` double backward for the gradient penalty) against which the explicit
    numpy formulas of ``np_ops``/``step`` are checked in ``tests/test_oracle.py``;
  * the ``cpu_baseline`` leg of ``bench.py`` (kind "port"): the closest available stand-in for the
    reference's TF/Keras-CPU path (same class of oneDNN/Eigen kernels + tape autodiff), timed on
    the GPU box's host cores.
It mirrors wgan.py:86-172, 234-285 literally (the [B]-vector loss is summed by autograd exactly
as ``tf.GradientTape.gradient`` does for a non-scalar target).
"""
from __future__ import annotations

import math
import numpy as np
import torch
import torch.nn.functional as F

from . import np_ops as O
from . import models as M


def _same_pad(x, k, s):
    """NCHW tensor -> explicitly, asymmetrically zero-padded for [TF] SAME."""
    H, W = x.shape[2], x.shape[3]
    _, pt, pb = O.same_pads(H, k, s)
    _, pl, pr = O.same_pads(W, k, s)
    return F.pad(x, (pl, pr, pt, pb))


def conv2d(x, w, stride):
    """x NHWC, w [kh,kw,ci,co] -> NHWC."""
    xn = x.permute(0, 3, 1, 2)
    y = F.conv2d(_same_pad(xn, w.shape[0], stride), w.permute(3, 2, 0, 1), stride=stride)
    return y.permute(0, 2, 3, 1)


def conv2d_transpose(x, w, stride):
    """Keras Conv2DTranspose 'same': w [kh,kw,c_out,c_in].  Full transposed conv, then crop by the
    SAME pads of the matching forward conv."""
    k = w.shape[0]
    xn = x.permute(0, 3, 1, 2)
    H, W = x.shape[1] * stride, x.shape[2] * stride
    _, pt, pb = O.same_pads(H, k, stride)
    _, pl, pr = O.same_pads(W, k, stride)
    full = F.conv_transpose2d(xn, w.permute(3, 2, 0, 1), stride=stride)   # size (h-1)s+k
    # pad the full result on the bottom/right if the crop window exceeds it (s>1, output_padding)
    need_h, need_w = pt + H, pl + W
    ph, pw = max(need_h - full.shape[2], 0), max(need_w - full.shape[3], 0)
    if ph or pw:
        full = F.pad(full, (0, pw, 0, ph))
    y = full[:, :, pt:pt + H, pl:pl + W]
    return y.permute(0, 2, 3, 1)


def blur(x, std):
    ks, s, _ = O.blur_policy(std, x.shape[1], x.shape[2])
    g = torch.from_numpy(O.gaussian_kernel_1d(s, ks, dtype=np.float64)).to(x.dtype)
    if x.dtype == torch.float32:
        g = torch.from_numpy(O.gaussian_kernel_1d(s, ks, dtype=np.float32))
    T, C = g.shape[0], x.shape[3]
    xn = x.permute(0, 3, 1, 2)
    kv = g.view(1, 1, T, 1).repeat(C, 1, 1, 1)
    kh = g.view(1, 1, 1, T).repeat(C, 1, 1, 1)
    y = F.conv2d(xn, kv, padding=(T // 2, 0), groups=C)
    y = F.conv2d(y, kh, padding=(0, T // 2), groups=C)
    return y.permute(0, 2, 3, 1)


def forward(spec, params, x, training, masks=None, bn_updates=None):
    mi = 0
    for i, (L, p) in enumerate(zip(spec, params)):
        t = L["type"]
        if t == "dense":
            x = x @ p["kernel"]
            if L["use_bias"]:
                x = x + p["bias"]
        elif t == "reshape":
            x = x.reshape((x.shape[0],) + tuple(L["shape"]))
        elif t == "flatten":
            x = x.reshape(x.shape[0], -1)
        elif t in ("conv", "convT"):
            x = conv2d(x, p["kernel"], L["stride"]) if t == "conv" else conv2d_transpose(x, p["kernel"], L["stride"])
            if L["use_bias"]:
                x = x + p["bias"]
            if L.get("activation") == "tanh":
                x = torch.tanh(x)
        elif t == "bn":
            if training:
                ax = tuple(range(x.ndim - 1))
                n = int(np.prod([x.shape[a] for a in ax]))
                mean = x.mean(ax)
                var = ((x - mean) ** 2).mean(ax)
                if bn_updates is not None:
                    vu = var * (n / max(n - 1, 1)) if x.ndim == 4 else var
                    bn_updates[i] = ((p["moving_mean"] * O.BN_MOMENTUM + mean * (1 - O.BN_MOMENTUM)).detach(),
                                     (p["moving_var"] * O.BN_MOMENTUM + vu * (1 - O.BN_MOMENTUM)).detach())
            else:
                mean, var = p["moving_mean"], p["moving_var"]
            x = p["gamma"] * (x - mean) / torch.sqrt(var + O.BN_EPS) + p["beta"]
        elif t == "lrelu":
            x = F.leaky_relu(x, O.LRELU_ALPHA)
        elif t == "dropout":
            if training:
                x = x * (1.0 / (1.0 - L["rate"])) * masks[mi].to(x.dtype)
            mi += 1
    return x


def to_torch(params, dtype, requires_grad=True):
    out = []
    for p in params:
        q = {}
        for k, v in p.items():
            tns = torch.from_numpy(np.asarray(v)).to(dtype).clone()
            if requires_grad and k in M.TRAINABLE:
                tns.requires_grad_(True)
            q[k] = tns
        out.append(q)
    return out


def trainables(params):
    return [(i, k, p[k]) for i, p in enumerate(params) for k in p if k in M.TRAINABLE]


def gradient_penalty(dspec, dparams, std, reals, fakes, alpha):
    """wgan.py:234-246."""
    B = reals.shape[0]
    xhat = (reals + alpha.view(B, 1, 1, 1) * (fakes - reals)).detach().requires_grad_(True)
    yhat = forward(dspec, dparams, blur(xhat, std), training=False)
    grad, = torch.autograd.grad(yhat.sum(), xhat, create_graph=True)
    norm = grad.reshape(B, -1).norm(dim=1)
    return ((norm - 1.0) ** 2).mean()


def discriminator_step_grads(st_t, reals, rnd, hp):
    """wgan.py:132-151, 272-285 via autograd.  ``st_t``: dict(gspec,dspec,g,d,std) of torch params."""
    with torch.no_grad():
        fakes = forward(st_t["gspec"], st_t["g"], rnd["z_d"], training=False)
    fs = forward(st_t["dspec"], st_t["d"], blur(fakes, st_t["std"]), True, rnd["mask_fake"])
    rs = forward(st_t["dspec"], st_t["d"], blur(reals, st_t["std"]), True, rnd["mask_real"])
    loss = (fs - rs).sum() * (1.0 / hp["global_batch_size"])
    gp = gradient_penalty(st_t["dspec"], st_t["d"], st_t["std"], reals, fakes, rnd["alpha"])
    gp_term = hp["gp_coefficient"] * gp
    norm_term = hp["e_drift"] * (fs.norm(dim=-1) + rs.norm(dim=-1))
    disc_loss = loss + gp_term + norm_term            # shape [B]
    tr = trainables(st_t["d"])
    grads = torch.autograd.grad(disc_loss.sum(), [t for _, _, t in tr])
    out = [dict() for _ in st_t["d"]]
    for (i, k, _), g in zip(tr, grads):
        out[i][k] = g
    met = dict(fake_scores=float(fs.detach().mean()), real_scores=float(rs.detach().mean()), disc_loss=float(disc_loss.detach().mean()),
               gp_term=float(gp_term.detach()), norm_term=float(norm_term.detach().mean()))
    return out, met, fakes


def generator_step_grads(st_t, rnd, hp):
    """wgan.py:159-172 via autograd."""
    upd = {}
    fakes = forward(st_t["gspec"], st_t["g"], rnd["z_g"], training=True, bn_updates=upd)
    sc = forward(st_t["dspec"], st_t["d"], blur(fakes, st_t["std"]), training=False)
    gen_loss = -sc.sum() * (1.0 / hp["global_batch_size"])
    tr = trainables(st_t["g"])
    grads = torch.autograd.grad(gen_loss, [t for _, _, t in tr])
    out = [dict() for _ in st_t["g"]]
    for (i, k, _), g in zip(tr, grads):
        out[i][k] = g
    return out, upd, dict(gen_loss=float(gen_loss.detach()), fake_scores_g=float(sc.detach().mean()))


class TorchTrainer:
    """Stateful CPU trainer used for the ``cpu_baseline`` timing and for step-level cross-checks."""

    def __init__(self, arch, seed=0, dtype=torch.float32, std=0.05, hp=None):
        from .step import DEFAULT_HP
        rng = np.random.default_rng(seed)
        self.arch, self.dtype = arch, dtype
        self.hp = dict(DEFAULT_HP, **(hp or {}))
        self.gspec, self.dspec = M.generator_spec(arch), M.discriminator_spec(arch)
        self.g = to_torch(M.init_params(self.gspec, (M.LATENT[arch],), rng), dtype)
        self.d = to_torch(M.init_params(self.dspec, M.image_shape(arch), rng), dtype)
        self.std = std
        self.opt_g = torch.optim.Adam([t for _, _, t in trainables(self.g)], lr=self.hp["learning_rate"], eps=O.ADAM_EPS)
        self.opt_d = torch.optim.Adam([t for _, _, t in trainables(self.d)], lr=self.hp["learning_rate"], eps=O.ADAM_EPS)
        self.n_batches = 0

    def state(self):
        return dict(gspec=self.gspec, dspec=self.dspec, g=self.g, d=self.d, std=self.std)

    def draw(self, batch, gen):
        shp = [(batch,) + s for L, s in zip(self.dspec, M.infer_shapes(self.dspec, M.image_shape(self.arch)))
               if L["type"] == "dropout"]
        u = lambda *s: torch.rand(*s, generator=gen, dtype=self.dtype)
        mk = lambda: [(torch.rand(*s, generator=gen) >= 0.3) for s in shp]
        return dict(z_d=u(batch, M.LATENT[self.arch]), z_g=u(batch, M.LATENT[self.arch]), alpha=u(batch),
                    mask_fake=mk(), mask_real=mk())

    def train_on_batch(self, reals, rnd):
        st = self.state()
        dg, met, _ = discriminator_step_grads(st, reals, rnd, self.hp)
        for (i, k, t) in trainables(self.d):
            t.grad = dg[i][k]
        self.opt_d.step()
        if self.n_batches % self.hp["d_steps_per_g_step"] == 0:
            gg, upd, gm = generator_step_grads(st, rnd, self.hp)
            for (i, k, t) in trainables(self.g):
                t.grad = gg[i][k]
            self.opt_g.step()
            for i, (nm, nv) in upd.items():
                self.g[i]["moving_mean"], self.g[i]["moving_var"] = nm, nv
            met.update(gm)
        self.n_batches += 1
        return met
