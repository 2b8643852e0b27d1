"""One full ``train_on_batch`` of BlurredWGANGP in numpy, randomness injected.

Oracle = test infrastructure (see ``oracle/__init__.py``); parity unpinned.

Follows wgan.py:86-172 (step order, training flags), wgan.py:234-285 (GP + losses),
blurred_gan.py:29-48 (blur inside the critic) with the quirks of SURVEY.md 8a reproduced:
  Q1  disc_loss is a [B] vector; its gradient is that of its SUM           (wgan.py:279-285)
  Q2  losses are scaled by 1/hp.global_batch_size, whatever the real batch (wgan.py:130,157)
  Q4  the generator runs with training=False (moving BN stats) in the D-step (wgan.py:135)
  Q6  fake_scores metric averages the D-step and G-step fake scores        (wgan.py:143,170)
The GP second order uses the closed form of SURVEY.md 8a (one linearised forward + one wgrad per
layer); ``tests/test_oracle.py`` checks it against torch autograd double-backward.
"""
from __future__ import annotations

import copy
import numpy as np
from . import np_ops as O
from . import models as M


DEFAULT_HP = dict(learning_rate=0.001, d_steps_per_g_step=1, global_batch_size=32,
                  e_drift=1e-4, gp_coefficient=10.0)


def new_state(arch, rng, dtype=np.float32, std=0.05):
    gs, ds = M.generator_spec(arch), M.discriminator_spec(arch)
    gp = M.init_params(gs, (M.LATENT[arch],), rng, dtype)
    dp = M.init_params(ds, M.image_shape(arch), rng, dtype)
    zeros = lambda ps: [{k: np.zeros_like(v) for k, v in p.items() if k in M.TRAINABLE} for p in ps]
    return dict(arch=arch, gspec=gs, dspec=ds, g=gp, d=dp,
                g_m=zeros(gp), g_v=zeros(gp), d_m=zeros(dp), d_v=zeros(dp),
                g_t=0, d_t=0, n_img=0, n_batches=0, std=float(std))


def dropout_mask_shapes(arch, batch):
    ds = M.discriminator_spec(arch)
    shapes = M.infer_shapes(ds, M.image_shape(arch))
    return [(batch,) + s for L, s in zip(ds, shapes) if L["type"] == "dropout"]


def draw_randomness(arch, batch, rng, dtype=np.float32):
    """The random inputs of one step (wgan.py:118,237; Dropout masks), as explicit arrays."""
    shp = dropout_mask_shapes(arch, batch)
    mk = lambda: [(rng.uniform(size=s) >= 0.3).astype(np.uint8) for s in shp]
    return dict(z_d=rng.uniform(size=(batch, M.LATENT[arch])).astype(dtype),
                z_g=rng.uniform(size=(batch, M.LATENT[arch])).astype(dtype),
                alpha=rng.uniform(size=(batch,)).astype(dtype),
                mask_fake=mk(), mask_real=mk())


def critic_fwd(st, x, training, masks=None, lrelu_masks=None):
    """D~ = Sequential([GaussianBlur2D, D]) (blurred_gan.py:30-34)."""
    a0 = O.blur_images(x, st["std"])
    y, cache = M.forward(st["dspec"], st["d"], a0, training, masks, lrelu_masks=lrelu_masks)
    return y, cache


def gradient_penalty(st, reals, fakes, alpha, want_grads=True, lrelu_masks=None):
    """wgan.py:234-246 + its gradient w.r.t. critic weights (second order)."""
    B = reals.shape[0]
    a = alpha.reshape(B, 1, 1, 1).astype(reals.dtype)
    xhat = reals + a * (fakes - reals)
    yhat, cache = critic_fwd(st, xhat, training=False, lrelu_masks=lrelu_masks)
    dz = {}
    _, d0 = M.backward(st["dspec"], st["d"], cache, np.ones_like(yhat), need_dx=True, need_dw=False,
                       training=False, keep_dz=dz)
    ks, s, _ = O.blur_policy(st["std"], xhat.shape[1], xhat.shape[2])
    g = O.gaussian_blur(d0, s, ks)                       # blur is self-adjoint
    n = np.sqrt((g.reshape(B, -1) ** 2).sum(1))
    gp = ((n - 1) ** 2).mean()
    if not want_grads:
        return gp, None, n
    gbar = ((2.0 / B) * (n - 1) / n).astype(g.dtype).reshape(B, 1, 1, 1) * g
    v0 = O.gaussian_blur(gbar, s, ks)
    grads = M.linearised_forward(st["dspec"], st["d"], cache, dz, v0)
    return gp, grads, n


def _acc(dst, src, scale=1.0):
    for gd, gs in zip(dst, src):
        for k, v in gs.items():
            gd[k] = gd.get(k, 0) + scale * v
    return dst


def discriminator_grads(st, reals, rnd, hp, force=None):
    """wgan.py:132-151 + 272-285.  Returns (grads, metrics, fakes).  ``force`` (test aid, see models.forward): LeakyReLU
    branches per critic pass, {"fake": [...], "real": [...], "hat": [...]}."""
    force = force or {}
    B = reals.shape[0]
    dt = reals.dtype.type
    inv_gbs = dt(1.0 / hp["global_batch_size"])
    # data parallel (build-side definition): this call sees one shard of a global batch of
    # B * dp_world samples; its gradients are meant to be SUMMED over the shards.
    Bg = B * int(hp.get("dp_world", 1))
    fakes, _ = M.forward(st["gspec"], st["g"], rnd["z_d"], training=False)           # Q4
    fs, cf = critic_fwd(st, fakes, True, rnd["mask_fake"], force.get("fake"))
    rs, cr = critic_fwd(st, reals, True, rnd["mask_real"], force.get("real"))
    l_w = (fs - rs).sum() * inv_gbs
    gp, gp_grads, _ = gradient_penalty(st, reals, fakes, rnd["alpha"], lrelu_masks=force.get("hat"))
    gp_term = dt(hp["gp_coefficient"]) * gp
    norm_term = dt(hp["e_drift"]) * (np.abs(fs[:, 0]) + np.abs(rs[:, 0]))             # [B]
    disc_loss_vec = l_w + gp_term + norm_term                                           # Q1: [B]
    # gradient of sum(disc_loss_vec): B*(l_w + gp_term) + sum(norm_term)   (B = the global batch under DP)
    dfs = (Bg * inv_gbs + dt(hp["e_drift"]) * np.sign(fs)).astype(reals.dtype)
    drs = (-Bg * inv_gbs + dt(hp["e_drift"]) * np.sign(rs)).astype(reals.dtype)
    gf, _ = M.backward(st["dspec"], st["d"], cf, dfs, need_dx=False)
    gr, _ = M.backward(st["dspec"], st["d"], cr, drs, need_dx=False)
    grads = [dict() for _ in st["dspec"]]
    _acc(grads, gf)
    _acc(grads, gr)
    # gp_grads = d/dW mean_local((n-1)^2); global mean = (1/dp_world) * sum of local means, times the Q1 factor Bg
    _acc(grads, gp_grads, scale=dt(Bg * hp["gp_coefficient"] / int(hp.get("dp_world", 1))))
    metrics = dict(fake_scores=float(fs.mean()), real_scores=float(rs.mean()),
                   disc_loss=float(disc_loss_vec.mean()), gp_term=float(gp_term),
                   norm_term=float(norm_term.mean()))
    return grads, metrics, fakes


def generator_grads(st, rnd, hp, batch, force=None):
    """wgan.py:159-172.  ``force`` (test aid): LeakyReLU branches, {"g": [...] generator, "d_gstep": [...] critic}."""
    force = force or {}
    upd = {}
    fakes, cg = M.forward(st["gspec"], st["g"], rnd["z_g"], training=True, update_bn=upd, lrelu_masks=force.get("g"))
    ks, s, _ = O.blur_policy(st["std"], fakes.shape[1], fakes.shape[2])
    a0 = O.gaussian_blur(fakes, s, ks)
    sc, cd = M.forward(st["dspec"], st["d"], a0, training=False, lrelu_masks=force.get("d_gstep"))
    inv_gbs = fakes.dtype.type(1.0 / hp["global_batch_size"])
    gen_loss = -sc.sum() * inv_gbs
    dsc = np.full_like(sc, -inv_gbs)
    _, da0 = M.backward(st["dspec"], st["d"], cd, dsc, need_dx=True, need_dw=False, training=False)
    dfakes = O.gaussian_blur(da0, s, ks)
    grads, _ = M.backward(st["gspec"], st["g"], cg, dfakes, need_dx=False, need_dw=True, training=True)
    return grads, upd, dict(gen_loss=float(gen_loss), fake_scores_g=float(sc.mean()))


def _adam(params, ms, vs, grads, t, lr):
    for p, m, v, g in zip(params, ms, vs, grads):
        for k in g:
            p[k], m[k], v[k] = O.adam_update(p[k], m[k], v[k], g[k].astype(p[k].dtype), t, lr)


def train_on_batch(st, reals, rnd, hp=None, force=None):
    """wgan.py:86-114.  Mutates and returns ``st``; also returns the metrics dict and aux grads.  ``force``: see
    discriminator_grads / generator_grads (test aid)."""
    hp = dict(DEFAULT_HP, **(hp or {}))
    B = reals.shape[0]
    dg, met, fakes = discriminator_grads(st, reals, rnd, hp, force)
    st["d_t"] += 1
    _adam(st["d"], st["d_m"], st["d_v"], dg, st["d_t"], hp["learning_rate"])
    aux = dict(d_grads=dg, fakes=fakes)
    if st["n_batches"] % hp["d_steps_per_g_step"] == 0:
        gg, upd, gm = generator_grads(st, rnd, hp, B, force)
        st["g_t"] += 1
        _adam(st["g"], st["g_m"], st["g_v"], gg, st["g_t"], hp["learning_rate"])
        for i, (nm, nv) in upd.items():
            st["g"][i]["moving_mean"], st["g"][i]["moving_var"] = nm, nv
        met["gen_loss"] = gm["gen_loss"]
        met["fake_scores"] = 0.5 * (met["fake_scores"] + gm["fake_scores_g"])             # Q6
        aux["g_grads"] = gg
    met["std"] = st["std"]
    st["n_img"] += B
    st["n_batches"] += 1
    return st, met, aux


def clone_state(st):
    return copy.deepcopy(st)
