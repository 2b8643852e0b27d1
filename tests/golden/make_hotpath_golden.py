"""Generates the frozen hot-path fixtures under tests/golden/ from the float64 oracle.

    python tests/golden/make_hotpath_golden.py            # writes hotpath_ops.npz, hotpath_step_tiny.npz, hotpath_step_mnist.npz

What the files are for: every GPU test recomputes the oracle on the fly, so an edit that moves oracle and kernels together
would be invisible.  These files freeze inputs AND expected outputs; ``tests/test_golden_cpu.py`` checks today's oracle
against them (oracle drift), ``tests/test_golden_gpu.py`` checks the HIP path against them (no oracle import at all).
The oracle restates /root/reference/gaussian_blur.py:83-132 and wgan.py:132-172,234-285 (parity unpinned upstream:
the reference holds no vectors and TensorFlow is absent, see oracle/__init__.py); regenerate ONLY when the oracle is
deliberately changed, and say so in the commit.

Inputs are float32-representable values; expected outputs are the oracle's float64 results on them.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, HERE)

from oracle import np_ops as O      # noqa: E402
from oracle import models as M      # noqa: E402
from oracle import step as S        # noqa: E402
import hashgen as Hg                # noqa: E402

WKEYS = ("kernel", "bias", "gamma", "beta", "moving_mean", "moving_var")
GKEYS = ("kernel", "bias", "gamma", "beta")
SAMPLE = 4096


def f32(a):
    """float64 array holding float32-representable values."""
    return np.asarray(a, dtype=np.float32).astype(np.float64)


def make_ops():
    rng = np.random.default_rng(20261004)
    out = {}
    # ---- blur (gaussian_blur.py:50-132): 3 / 31 / 143 taps + a 1-channel and a clipped case
    blur_cases = [("blur3", (2, 16, 16, 3), 0.05), ("blur31", (2, 32, 32, 3), 5.0), ("blur143", (1, 144, 16, 3), 23.5),
                  ("blur13_c1", (2, 28, 28, 1), 2.0), ("blur_clip29", (1, 28, 28, 1), 9.0), ("blur31_128", (1, 128, 128, 3), 4.94)]
    out["blur_names"] = np.array([c[0] for c in blur_cases])
    for name, shape, sigma in blur_cases:
        x = f32(rng.uniform(-1, 1, size=shape))
        ks, se, nt = O.blur_policy(sigma, shape[1], shape[2])
        out[name + "_x"] = x.astype(np.float32)
        out[name + "_sigma"] = np.float64(sigma)
        out[name + "_policy"] = np.array([ks, se, nt], dtype=np.float64)
        out[name + "_taps"] = O.gaussian_kernel_1d(se, ks, dtype=np.float64)
        out[name + "_taps32"] = O.gaussian_kernel_1d(se, ks, dtype=np.float32)
        out[name + "_y"] = O.blur_images(x, sigma)
    # ---- conv family (T1 / T2 and their tape gradients): one thin-K, one thin-N and two MFMA shapes
    conv_cases = [("conv_thin_k", (2, 16, 16, 3, 32, 2)), ("conv_thin_n", (2, 16, 16, 32, 3, 1)),
                  ("conv_mfma_s2", (2, 8, 8, 32, 64, 2)), ("conv_mfma_s1", (3, 4, 4, 64, 32, 1))]
    out["conv_names"] = np.array([c[0] for c in conv_cases])
    for name, (B, H, W, Ci, Co, s) in conv_cases:
        x = f32(rng.uniform(-1, 1, size=(B, H, W, Ci)))
        w = f32(rng.uniform(-1, 1, size=(5, 5, Ci, Co)) / np.sqrt(25 * Ci))
        dy = f32(rng.uniform(-1, 1, size=(B, -(-H // s), -(-W // s), Co)))
        out[name + "_geom"] = np.array([B, H, W, Ci, Co, s])
        out[name + "_x"], out[name + "_w"], out[name + "_dy"] = x.astype(np.float32), w.astype(np.float32), dy.astype(np.float32)
        out[name + "_y"] = O.conv2d_fwd(x, w, s)
        out[name + "_dx"] = O.conv2d_bwd_data(dy, w, s, (H, W))
        out[name + "_dw"] = O.conv2d_bwd_filter(x, dy, s, 5)
    # ---- BatchNorm + LeakyReLU, training forward and backward (T5 / T6): 4-D (unbiased moving rule) and 2-D (biased)
    out["bn_names"] = np.array(["bn4d", "bn2d"])
    for name, shape in (("bn4d", (4, 4, 4, 32)), ("bn2d", (16, 40))):
        C = shape[-1]
        x = f32(rng.normal(size=shape) * 2 + 0.5)
        gamma, beta = f32(1 + 0.3 * rng.normal(size=C)), f32(0.2 * rng.normal(size=C))
        mm, mv = f32(rng.normal(size=C)), f32(1 + rng.uniform(size=C))
        dy = f32(rng.normal(size=shape))
        u, cache, nm, nv = O.bn_train_fwd(x, gamma, beta, mm, mv)
        dx, dg, db = O.bn_train_bwd(dy * O.lrelu_mask(u), gamma, cache)
        for k, v in dict(x=x, gamma=gamma, beta=beta, mm=mm, mv=mv, dy=dy).items():
            out[f"{name}_{k}"] = v.astype(np.float32)
        out[name + "_y"], out[name + "_new_mm"], out[name + "_new_mv"] = O.lrelu_fwd(u), nm, nv
        out[name + "_y_infer"] = O.lrelu_fwd(O.bn_infer_fwd(x, gamma, beta, mm, mv))
        out[name + "_dx"], out[name + "_dgamma"], out[name + "_dbeta"] = dx, dg, db
    # ---- Adam (T9), t = 1 and t = 2 from zero slots
    n = 1000
    th, g1, g2 = f32(rng.normal(size=n)), f32(rng.normal(size=n) * 10.0 ** rng.integers(-6, 1, size=n)), f32(rng.normal(size=n) * 1e-3)
    t1 = O.adam_update(th, np.zeros(n), np.zeros(n), g1, 1, 1e-3)
    t2 = O.adam_update(*t1, g2, 2, 1e-3)
    out.update(adam_theta=th.astype(np.float32), adam_g1=g1.astype(np.float32), adam_g2=g2.astype(np.float32),
               adam_theta1=t1[0], adam_m1=t1[1], adam_v1=t1[2], adam_theta2=t2[0], adam_m2=t2[1], adam_v2=t2[2])
    # ---- Dense (T4), loss (wgan.py:130,277-285 incl. Q1), x-hat / per-sample norm / second-order seed (wgan.py:237-246)
    A, Bm, bias = f32(rng.normal(size=(6, 100))), f32(rng.normal(size=(100, 48)) * 0.1), f32(rng.normal(size=48))
    out.update(dense_x=A.astype(np.float32), dense_w=Bm.astype(np.float32), dense_b=bias.astype(np.float32), dense_y=A @ Bm + bias)
    Bn = 6
    fs, rs, norms = f32(rng.normal(size=Bn)), f32(rng.normal(size=Bn)), f32(1 + rng.uniform(size=Bn))
    gp = ((norms - 1) ** 2).mean()
    nt = 1e-4 * (np.abs(fs) + np.abs(rs))
    out.update(loss_fs=fs.astype(np.float32), loss_rs=rs.astype(np.float32), loss_norms=norms.astype(np.float32),
               loss_metrics=np.array([fs.mean(), rs.mean(), (fs - rs).sum() / 32 + 10 * gp + nt.mean(), 10 * gp, nt.mean(), gp]),
               loss_dfs=Bn / 32 + 1e-4 * np.sign(fs), loss_drs=-Bn / 32 + 1e-4 * np.sign(rs),
               gloss_metrics=np.array([fs.mean(), -fs.sum() / 32]))
    r, f, a = f32(rng.normal(size=(Bn, 192))), f32(rng.normal(size=(Bn, 192))), f32(rng.uniform(size=Bn))
    nn = np.linalg.norm(f, axis=1)
    out.update(gp_r=r.astype(np.float32), gp_f=f.astype(np.float32), gp_a=a.astype(np.float32), gp_xhat=r + a[:, None] * (f - r),
               gp_norm=nn, gp_seed=0.7 * ((nn - 1) / nn)[:, None] * f)
    np.savez_compressed(os.path.join(HERE, "hotpath_ops.npz"), **out)
    return out


def flat_weights(params, keys=WKEYS):
    return [np.asarray(p[k]) for p in params for k in keys if k in p]


def make_step(arch, B, sigma, seed, steps, full):
    """One (or more) full train_on_batch calls (wgan.py:86-114) with every random input stored."""
    rng = np.random.default_rng(seed)
    st = S.new_state(arch, rng, np.float64, std=sigma)
    (st["g"], g_kinds), (st["d"], d_kinds) = Hg.hashed_params(st["g"], 1000 * seed), Hg.hashed_params(st["d"], 1000 * seed + 500)
    hp = dict(S.DEFAULT_HP, global_batch_size=B + 1)          # != B: the 1/global_batch_size scale (Q2) is visible
    out = dict(arch=np.array(arch), B=np.int64(B), sigma=np.float64(sigma), steps=np.int64(steps), hash_seed=np.int64(seed),
               hp_names=np.array(sorted(hp)), hp_values=np.array([hp[k] for k in sorted(hp)], dtype=np.float64),
               g_kinds=np.array(g_kinds), d_kinds=np.array(d_kinds), g_hash_seed=np.int64(1000 * seed), d_hash_seed=np.int64(1000 * seed + 500))
    for key in ("g", "d"):
        ws = flat_weights(st[key])
        out[f"{key}_nvars"] = np.int64(len(ws))
        for i, w in enumerate(ws):
            out[f"{key}_w{i:02d}_check"] = Hg.checksum(w)
            out[f"{key}_w{i:02d}_shape"] = np.array(w.shape)
            if full:
                out[f"{key}_w{i:02d}"] = w.astype(np.float32)
    H, W, C = M.image_shape(arch)
    for it in range(steps):
        reals = f32(rng.uniform(-1, 1, size=(B, H, W, C)))
        rnd = S.draw_randomness(arch, B, rng, np.float32)
        rnd = {k: (v.astype(np.float64) if isinstance(v, np.ndarray) else v) for k, v in rnd.items()}   # masks stay uint8 lists
        out[f"s{it}_reals"] = reals.astype(np.float32)
        for k in ("z_d", "z_g", "alpha"):
            out[f"s{it}_{k}"] = rnd[k].astype(np.float32)
        for k in ("mask_fake", "mask_real"):
            for j, m in enumerate(rnd[k]):
                out[f"s{it}_{k}{j}"] = np.packbits(m.astype(np.uint8).ravel())
                out[f"s{it}_{k}{j}_shape"] = np.array(m.shape)
        st, met, aux = S.train_on_batch(st, reals, rnd, hp)
        out[f"s{it}_metric_names"] = np.array(sorted(met))
        out[f"s{it}_metrics"] = np.array([met[k] for k in sorted(met)], dtype=np.float64)
        out[f"s{it}_fakes_check"] = Hg.checksum(aux["fakes"])
        out[f"s{it}_fakes_head"] = aux["fakes"].ravel()[:512]
        for key, grads in (("g", aux["g_grads"]), ("d", aux["d_grads"])):
            gl = [np.asarray(g[k]) for g in grads for k in GKEYS if k in g]
            wl = flat_weights(st[key])
            for i, g in enumerate(gl):
                idx = Hg.hashed_indices(77 + i, g.size, SAMPLE) if not full else np.arange(g.size)
                out[f"s{it}_{key}_grad{i:02d}_norm"] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                out[f"s{it}_{key}_grad{i:02d}_max"] = np.float64(np.abs(g).max())
                out[f"s{it}_{key}_grad{i:02d}"] = g.ravel()[idx]
            for i, w in enumerate(wl):
                idx = Hg.hashed_indices(77 + i, w.size, SAMPLE) if not full else np.arange(w.size)
                out[f"s{it}_{key}_after{i:02d}"] = w.ravel()[idx]
                out[f"s{it}_{key}_after{i:02d}_check"] = Hg.checksum(w)
    path = os.path.join(HERE, f"hotpath_step_{arch}.npz")
    np.savez_compressed(path, **out)
    return path


if __name__ == "__main__":
    make_ops()
    print(make_step("tiny", 3, 0.9, 11, steps=2, full=True))
    print(make_step("mnist", 3, 0.05, 12, steps=1, full=False))
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))
