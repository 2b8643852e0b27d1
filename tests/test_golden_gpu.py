"""GPU parity against the FROZEN fixtures of tests/golden/ (inputs and expected outputs in the files; the oracle is not
imported here -- tests/test_golden_cpu.py checks the oracle against the same files).  Covers the op set of
/root/reference/gaussian_blur.py:83-132 and wgan.py:132-172,234-285: the blur at 3 / 31 / 143 taps, conv forward / data
gradient / filter gradient at thin and MFMA shapes, BatchNorm training forward + backward, Adam, Dense, the losses, and whole
train_on_batch calls of the `tiny` and `mnist` stacks with their randomness."""
import math

import numpy as np
import pytest
import torch

import golden_io as G
from helpers import dev, conv_tol, product_grads, product_slots

pytestmark = pytest.mark.gpu


def test_blur_fixtures():
    from blurred_gan_amd import ops
    z = G.load_ops()
    for name in z["blur_names"]:
        x, sigma = z[f"{name}_x"], float(z[f"{name}_sigma"])
        B, H, W, C = x.shape
        ks, se, nt = ops.blur_policy(sigma, H, W)
        assert [ks, nt] == [z[f"{name}_policy"][0], z[f"{name}_policy"][2]] and abs(se - z[f"{name}_policy"][1]) < 1e-6
        taps = ops.gauss_kernel_1d(se, ks)
        np.testing.assert_allclose(taps, z[f"{name}_taps"], rtol=2e-6, atol=1e-9)
        nb = ops.blur_workspace_bytes(B, H, W, C, nt)
        tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None
        y = ops.blur_nhwc(dev(x), torch.empty(x.shape, device="cuda"), dev(np.asarray(taps)), nt, tmp)
        torch.cuda.synchronize()
        np.testing.assert_allclose(y.cpu().numpy(), z[f"{name}_y"], rtol=1e-5, atol=4e-6, err_msg=name)
        assert G.rel_l2(y.cpu().numpy(), z[f"{name}_y"]) < 2e-6, name


def test_conv_fixtures():
    from blurred_gan_amd import ops
    z = G.load_ops()
    for name in z["conv_names"]:
        B, H, W, Ci, Co, s = (int(v) for v in z[f"{name}_geom"])
        x, w, dy = z[f"{name}_x"], z[f"{name}_w"], z[f"{name}_dy"]
        wT = dev(np.transpose(w, (0, 1, 3, 2)))
        y = ops.conv2d_fwd(dev(x), wT, torch.empty(z[f"{name}_y"].shape, device="cuda"), 5, s)
        dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s)
        nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, s)
        ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
        dw = ops.conv2d_bwd_filter(dev(x), dev(dy), torch.full(w.shape, 7.0, device="cuda"), 5, s, 0.0, 1.0, ws)
        torch.cuda.synchronize()
        for got, key, K in ((y, "y", 25 * Ci), (dx, "dx", 25 * Co), (dw, "dw", dy.size // Co)):
            ref = z[f"{name}_{key}"]
            np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=1e-4, atol=conv_tol(K, np.abs(ref).max()), err_msg=f"{name} {key}")
            assert G.rel_l2(got.cpu().numpy(), ref) < 2e-6, (name, key)


def test_batchnorm_fixtures():
    from blurred_gan_amd import ops
    z = G.load_ops()
    for name in z["bn_names"]:
        x = z[f"{name}_x"]
        shape, C = x.shape, x.shape[-1]
        M = x.size // C
        ws = torch.empty(ops._lib.load().bg_bn_workspace_bytes(M, C) // 4 + 4, device="cuda")
        gamma, beta, mm, mv = (dev(z[f"{name}_{k}"]) for k in ("gamma", "beta", "mm", "mv"))
        y, sm, si = torch.empty(shape, device="cuda"), torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        yi = ops.bn_infer_fwd(dev(x), torch.empty(shape, device="cuda"), M, C, gamma, beta, mm, mv)
        np.testing.assert_allclose(yi.cpu().numpy(), z[f"{name}_y_infer"], rtol=1e-4, atol=2e-5)
        ops.bn_train_fwd(dev(x), y, M, C, gamma, beta, mm, mv, sm, si, ws, unbiased=(len(shape) == 4))
        np.testing.assert_allclose(y.cpu().numpy(), z[f"{name}_y"], rtol=1e-4, atol=2e-5)
        np.testing.assert_allclose(mm.cpu().numpy(), z[f"{name}_new_mm"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(mv.cpu().numpy(), z[f"{name}_new_mv"], rtol=1e-4, atol=1e-5)
        dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
        dx = ops.bn_train_bwd(dev(z[f"{name}_dy"]), y, dev(x), torch.empty(shape, device="cuda"), M, C, gamma, sm, si, dg, db, ws)
        np.testing.assert_allclose(dg.cpu().numpy(), z[f"{name}_dgamma"], rtol=2e-4, atol=2e-4 * math.sqrt(M))
        np.testing.assert_allclose(db.cpu().numpy(), z[f"{name}_dbeta"], rtol=2e-4, atol=2e-4 * math.sqrt(M))
        np.testing.assert_allclose(dx.cpu().numpy(), z[f"{name}_dx"], rtol=2e-4, atol=2e-5 * max(1, np.abs(z[f"{name}_dx"]).max()))
        assert G.rel_l2(dx.cpu().numpy(), z[f"{name}_dx"]) < 1e-5 and G.rel_l2(y.cpu().numpy(), z[f"{name}_y"]) < 2e-6


def test_adam_dense_loss_fixtures():
    from blurred_gan_amd import ops
    z = G.load_ops()
    n = z["adam_theta"].size
    th, m, v = dev(z["adam_theta"]), torch.zeros(n, device="cuda"), torch.zeros(n, device="cuda")
    for t, g in ((1, "adam_g1"), (2, "adam_g2")):
        ops.adam(th, m, v, dev(z[g]), 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t))
        np.testing.assert_allclose(th.cpu().numpy(), z[f"adam_theta{t}"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(m.cpu().numpy(), z[f"adam_m{t}"], rtol=1e-5, atol=2e-7)
        np.testing.assert_allclose(v.cpu().numpy(), z[f"adam_v{t}"], rtol=3e-5, atol=1e-12)
    M, K = z["dense_x"].shape
    N = z["dense_w"].shape[1]
    y = ops.gemm(dev(z["dense_x"]), dev(z["dense_w"]), torch.empty(M, N, device="cuda"), M, N, K, bias=dev(z["dense_b"]))
    np.testing.assert_allclose(y.cpu().numpy(), z["dense_y"], rtol=1e-5, atol=1e-5)
    B = z["loss_fs"].size
    dfs, drs, met = torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.empty(8, device="cuda")
    ops.wgangp_d_loss(dev(z["loss_fs"]), dev(z["loss_rs"]), dev(z["loss_norms"]), 1 / 32, 10.0, 1e-4, float(B), dfs, drs, met)
    np.testing.assert_allclose(met.cpu().numpy()[:6], z["loss_metrics"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dfs.cpu().numpy(), z["loss_dfs"], rtol=1e-6)
    np.testing.assert_allclose(drs.cpu().numpy(), z["loss_drs"], rtol=1e-6)
    ds, gm = torch.empty(B, device="cuda"), torch.empty(4, device="cuda")
    ops.wgan_g_loss(dev(z["loss_fs"]), 1 / 32, ds, gm)
    np.testing.assert_allclose(gm.cpu().numpy()[:2], z["gloss_metrics"], rtol=1e-5, atol=1e-7)
    r, f, a = dev(z["gp_r"]), dev(z["gp_f"]), dev(z["gp_a"])
    np.testing.assert_allclose(ops.lerp(r, f, a, torch.empty_like(r)).cpu().numpy(), z["gp_xhat"], rtol=1e-5, atol=1e-6)
    nr = ops.row_norm(f, torch.empty(B, device="cuda"))
    np.testing.assert_allclose(nr.cpu().numpy(), z["gp_norm"], rtol=1e-5)
    np.testing.assert_allclose(ops.gp_seed(f, nr, 0.7, torch.empty_like(f)).cpu().numpy(), z["gp_seed"], rtol=1e-4, atol=1e-6)


@pytest.mark.parametrize("arch", ["tiny", "mnist"])
def test_train_on_batch_fixtures(arch):
    """Whole steps through the reference-shaped API (wgan.py:86-114): metrics, the gradients (read back as Adam's first
    moment after the very first update, m = 0.1 g) and every variable after each step, against the stored values."""
    import blurred_gan_amd as bg
    from blurred_gan_amd import models
    fx = G.StepFixture(arch)
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=fx.sigma, global_batch_size=fx.hp["global_batch_size"], batch_size=fx.B,
                                          learning_rate=fx.hp["learning_rate"], e_drift=fx.hp["e_drift"],
                                          gp_coefficient=fx.hp["gp_coefficient"], d_steps_per_g_step=fx.hp["d_steps_per_g_step"])
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir="/tmp/bg_test_logs"))
    for model, key in ((gen, "g"), (disc, "d")):
        model.build()
        model.set_weights(fx.weights(key))
    report = {}
    for it in range(fx.steps):
        got = dict(zip(gan.metrics_names, gan.train_on_batch(fx.reals(it), randomness=fx.randomness(it))))
        want = fx.metrics(it)
        for k in ("real_scores", "fake_scores", "disc_loss", "gp_term", "norm_term", "gen_loss", "std"):
            assert abs(got[k] - want[k]) < 1e-4 * max(1, abs(want[k])), (it, k, got[k], want[k])
        for model, key in ((gan.generator, "g"), (gan.discriminator, "d")):
            if it == 0:                                   # m_1 = (1 - beta1) * g
                for i, (m, (exp, norm, mx)) in enumerate(zip(product_slots(model, "m"), fx.grads(0, key))):
                    g = m.ravel().astype(np.float64) / (1.0 - np.float32(0.9).astype(np.float64))
                    gs = g[fx.sample_index(i, g.size)]
                    # elementwise: all but <= 0.5 % of the elements within rtol 2e-3 + 2e-4 (critic) / 2e-3 (generator) of the
                    # variable's maximum and none beyond 1 % of it (the generator's gradients come through BatchNorm backwards
                    # over a batch of 3: single elements are cancellation residues; measured worst 1.2e-3 of the maximum); the
                    # per-tensor relative L2 error and cosine below are the sharp criteria
                    err = np.abs(gs - exp)
                    atol = (2e-3 if key == "g" else 2e-4) * mx
                    assert (err > 2e-3 * np.abs(exp) + atol).mean() <= 0.005 and err.max() <= 1e-2 * mx, (key, i, err.max(), mx)
                    if not np.any(exp):                       # a gradient that is exactly zero (the critic's Dense bias here)
                        assert np.abs(gs).max() <= 1e-6, (key, i)
                        continue
                    l2, cs = G.rel_l2(gs, exp), G.cosine(gs, exp)
                    report[f"{key}{i:02d}"] = (l2, 1 - cs)
                    assert l2 < (1e-3 if key == "g" else 2e-4) and cs > 1 - 1e-6, (key, i, l2, cs)
                    assert abs(np.linalg.norm(g) - norm) < 1e-3 * norm
            tr = fx.trainable(key)
            for i, (w, (exp, chk)) in enumerate(zip(model.get_weights(), fx.after(it, key))):
                ws = w.ravel()[fx.sample_index(i, w.size)]
                if tr[i]:
                    # an Adam step is lr * m / (sqrt(v) + 1e-7): where |g| is float32 noise the update is not determined
                    # beyond +-lr, everywhere else it agrees to 1e-3 relative
                    assert np.abs(ws - exp).max() <= 2.0 * fx.hp["learning_rate"] * (it + 1) + 1e-6
                    assert (np.abs(ws - exp) > 1e-3 * np.abs(exp) + 1e-4).mean() <= 0.005, (key, i)
                else:
                    np.testing.assert_allclose(ws, exp, rtol=2e-4, atol=2e-6, err_msg=f"{key} variable {i}")
    assert int(gan.n_batches) == fx.steps and int(gan.n_img) == fx.steps * fx.B
    print(arch, "gradient (rel L2, 1 - cos) per variable:", {k: (f"{a:.1e}", f"{b:.1e}") for k, (a, b) in report.items()})
