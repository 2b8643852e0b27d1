"""Oracle drift detection: today's oracle/ against the FROZEN fixtures of tests/golden/ (inputs and expected outputs
stored; written once by tests/golden/make_hotpath_golden.py).  The GPU counterpart, tests/test_golden_gpu.py, checks the HIP
path against the same files without importing the oracle, so an edit that moves oracle and kernels together shows up here."""
import numpy as np
import pytest

from oracle import np_ops as O
from oracle import models as M
from oracle import step as S
import golden_io as G

TIGHT = dict(rtol=1e-11, atol=1e-13)


def test_blur_fixtures():
    z = G.load_ops()
    for name in z["blur_names"]:
        x, sigma = z[f"{name}_x"].astype(np.float64), float(z[f"{name}_sigma"])
        ks, se, nt = O.blur_policy(sigma, x.shape[1], x.shape[2])
        np.testing.assert_allclose([ks, se, nt], z[f"{name}_policy"], rtol=0, atol=0)
        np.testing.assert_allclose(O.gaussian_kernel_1d(se, ks, dtype=np.float64), z[f"{name}_taps"], **TIGHT)
        assert np.array_equal(O.gaussian_kernel_1d(se, ks, dtype=np.float32), z[f"{name}_taps32"])
        np.testing.assert_allclose(O.blur_images(x, sigma), z[f"{name}_y"], **TIGHT)
    assert [int(z[f"{n}_policy"][2]) for n in ("blur3", "blur31", "blur143", "blur_clip29", "blur31_128")] == [3, 31, 143, 29, 31]


def test_conv_fixtures():
    z = G.load_ops()
    for name in z["conv_names"]:
        B, H, W, Ci, Co, s = (int(v) for v in z[f"{name}_geom"])
        x, w, dy = (z[f"{name}_{k}"].astype(np.float64) for k in ("x", "w", "dy"))
        np.testing.assert_allclose(O.conv2d_fwd(x, w, s), z[f"{name}_y"], **TIGHT)
        np.testing.assert_allclose(O.conv2d_bwd_data(dy, w, s, (H, W)), z[f"{name}_dx"], **TIGHT)
        np.testing.assert_allclose(O.conv2d_bwd_filter(x, dy, s, 5), z[f"{name}_dw"], **TIGHT)
        # Conv2DTranspose forward IS the data gradient (demo_celeba.py:62-87)
        np.testing.assert_allclose(O.conv2d_transpose_fwd(dy, w, s), z[f"{name}_dx"], **TIGHT)


def test_batchnorm_adam_dense_loss_fixtures():
    z = G.load_ops()
    for name in z["bn_names"]:
        x, gamma, beta, mm, mv, dy = (z[f"{name}_{k}"].astype(np.float64) for k in ("x", "gamma", "beta", "mm", "mv", "dy"))
        u, cache, nm, nv = O.bn_train_fwd(x, gamma, beta, mm, mv)
        dx, dg, db = O.bn_train_bwd(dy * O.lrelu_mask(u), gamma, cache)
        for got, key in ((O.lrelu_fwd(u), "y"), (nm, "new_mm"), (nv, "new_mv"), (dx, "dx"), (dg, "dgamma"), (db, "dbeta"),
                         (O.lrelu_fwd(O.bn_infer_fwd(x, gamma, beta, mm, mv)), "y_infer")):
            np.testing.assert_allclose(got, z[f"{name}_{key}"], **TIGHT)
    th, g1, g2 = (z[k].astype(np.float64) for k in ("adam_theta", "adam_g1", "adam_g2"))
    t1 = O.adam_update(th, np.zeros_like(th), np.zeros_like(th), g1, 1, 1e-3)
    t2 = O.adam_update(*t1, g2, 2, 1e-3)
    for got, key in zip(t1 + t2, ("theta1", "m1", "v1", "theta2", "m2", "v2")):
        np.testing.assert_allclose(got, z["adam_" + key], **TIGHT)
    np.testing.assert_allclose(O.dense_fwd(z["dense_x"].astype(np.float64), z["dense_w"].astype(np.float64), z["dense_b"].astype(np.float64)),
                               z["dense_y"], **TIGHT)


def _oracle_state(fx):
    st = S.new_state(fx.arch, np.random.default_rng(0), np.float64, std=fx.sigma)
    for key in ("g", "d"):
        it = iter(fx.weights(key))
        for p in st[key]:
            for k in G.Hg.WKEYS:
                if k in p:
                    p[k] = next(it).reshape(p[k].shape)
    return st


@pytest.mark.parametrize("arch", ["tiny", "mnist"])
def test_train_on_batch_fixtures(arch):
    """wgan.py:86-114 end to end: metrics, every gradient and every variable after the step(s)."""
    fx = G.StepFixture(arch)
    st = _oracle_state(fx)
    for it in range(fx.steps):
        st, met, aux = S.train_on_batch(st, fx.reals(it).astype(np.float64), fx.randomness(it), fx.hp)
        want = fx.metrics(it)
        assert sorted(met) == sorted(want)
        for k in want:
            assert abs(met[k] - want[k]) <= 1e-10 * max(1.0, abs(want[k])), (it, k, met[k], want[k])
        for key, grads in (("g", aux["g_grads"]), ("d", aux["d_grads"])):
            gl = [np.asarray(g[k]) for g in grads for k in ("kernel", "bias", "gamma", "beta") if k in g]
            for i, (g, (exp, norm, mx)) in enumerate(zip(gl, fx.grads(it, key))):
                np.testing.assert_allclose(g.ravel()[fx.sample_index(i, g.size)], exp, rtol=1e-9, atol=1e-12 * max(mx, 1e-30))
                assert abs(np.linalg.norm(g.ravel()) - norm) <= 1e-9 * norm
            wl = [np.asarray(p[k]) for p in st[key] for k in G.Hg.WKEYS if k in p]
            for i, (w, (exp, chk)) in enumerate(zip(wl, fx.after(it, key))):
                np.testing.assert_allclose(w.ravel()[fx.sample_index(i, w.size)], exp, rtol=1e-9, atol=1e-12)
                np.testing.assert_allclose(G.Hg.checksum(w), chk, rtol=1e-9, atol=1e-9)


def test_hash_generator_known_answers():
    """The counter hash behind the MNIST fixture's weights (its outputs are also pinned by the checksums in the file)."""
    u = G.Hg.hashed_uniform(5, 4)
    np.testing.assert_allclose(u, [0.21659902083849025, 0.9794872211475327, 0.612636434623909, -0.2114143599368985], rtol=0, atol=1e-15)
