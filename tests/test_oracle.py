"""CPU tests pinning the oracle: analytic known answers + an independent torch-autograd implementation.

The reference has no tests/fixtures for this path so these are the only pins
("parity unpinned" upstream)."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ops as O
from oracle import models as M
from oracle import step as S
from oracle import torch_ref as T


# ---------------------------------------------------------------- blur policy / kernel known answers
@pytest.mark.parametrize("std,hw,exp", [
    (0.05, 28, (3.0, 1 / 3, 3)), (0.34, 28, (3.0, 1 / 3, 3)), (0.5, 64, (4.0, 0.5, 5)),
    (1.0, 64, (7.0, 1.0, 7)), (2.0, 64, (13.0, 2.0, 13)), (5.0, 64, (31.0, 5.0, 31)),
    (4.94, 64, (30.0, 29 / 6, 31)), (10.5, 64, (64.0, 10.5, 65)), (23.5, 256, (142.0, 23.5, 143)),
    (42.34, 256, (255.0, 254 / 6, 255)), (23.5, 28, (28.0, 4.5, 29)), (23.5, 64, (64.0, 10.5, 65)),
    (23.5, 128, (128.0, 127 / 6, 129)),
])
def test_blur_policy_table(std, hw, exp):
    ks, s, taps = O.blur_policy(std, hw, hw)
    assert ks == exp[0] and taps == exp[2]
    assert abs(s - exp[1]) < 1e-6 * max(1.0, exp[1])   # float32 policy arithmetic


def test_gaussian_kernel_known_values():
    g = O.gaussian_kernel_1d(1 / 3, 3.0)
    np.testing.assert_allclose(g, [0.010867546, 0.97826493, 0.010867546], rtol=2e-6)
    g = O.gaussian_kernel_1d(1.0, 7.0)
    assert g.shape == (7,) and abs(g[3] - 0.399050) < 2e-6
    g = O.gaussian_kernel_1d(5.0, 31.0)
    assert g.shape == (31,) and abs(g[15] - 0.079940) < 2e-6 and abs(g[0] - 8.881e-4) < 2e-6
    for std, ks in [(0.5, 4.0), (10.5, 64.0), (23.5, 142.0)]:
        g = O.gaussian_kernel_1d(std, ks, np.float64)
        assert abs(g.sum() - 1) < 1e-12 and g.shape[0] % 2 == 1
        np.testing.assert_allclose(g, g[::-1])


def test_blur_constant_image_border_darkening():
    x = np.ones((1, 12, 12, 3), np.float64)
    ks, s, t = O.blur_policy(1.0, 12, 12)
    g = O.gaussian_kernel_1d(s, ks, np.float64)
    y = O.gaussian_blur(x, s, ks)
    assert abs(y[0, 6, 6, 0] - 1.0) < 1e-12           # interior unchanged
    corner = g[t // 2:].sum() ** 2                     # truncated sums in both directions
    assert abs(y[0, 0, 0, 1] - corner) < 1e-12


def test_blur_self_adjoint_and_matches_torch():
    rng = np.random.default_rng(0)
    x, y = rng.normal(size=(2, 9, 11, 3)), rng.normal(size=(2, 9, 11, 3))
    for std in (0.2, 1.0, 4.0):
        bx, by = O.blur_images(x, std), O.blur_images(y, std)
        assert abs((bx * y).sum() - (x * by).sum()) < 1e-10
        np.testing.assert_allclose(bx, T.blur(torch.from_numpy(x), std).numpy(), atol=1e-12)


# ---------------------------------------------------------------- conv family vs torch, adjointness
@pytest.mark.parametrize("H,W,s", [(8, 8, 2), (7, 7, 2), (6, 9, 1), (14, 14, 2), (4, 4, 1), (2, 2, 2)])
def test_conv_family_vs_torch_and_adjoint(H, W, s):
    rng = np.random.default_rng(1)
    x = rng.normal(size=(2, H, W, 3))
    w = rng.normal(size=(5, 5, 3, 4))
    y = O.conv2d_fwd(x, w, s)
    yt = T.conv2d(torch.from_numpy(x), torch.from_numpy(w), s).numpy()
    np.testing.assert_allclose(y, yt, atol=1e-11)
    dy = rng.normal(size=y.shape)
    dx = O.conv2d_bwd_data(dy, w, s, (H, W))
    dw = O.conv2d_bwd_filter(x, dy, s, 5)
    assert abs((y * dy).sum() - (x * dx).sum()) < 1e-9            # <conv x, dy> = <x, conv^T dy>
    assert abs((y * dy).sum() - (w * dw).sum()) < 1e-9
    # autograd agrees
    xt = torch.from_numpy(x).requires_grad_(True)
    wt = torch.from_numpy(w).requires_grad_(True)
    (T.conv2d(xt, wt, s) * torch.from_numpy(dy)).sum().backward()
    np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-10)
    np.testing.assert_allclose(dw, wt.grad.numpy(), atol=1e-10)


@pytest.mark.parametrize("h,s", [(4, 2), (7, 2), (4, 1), (7, 1), (3, 2)])
def test_conv_transpose_geometry_and_torch(h, s):
    """Pins the reference's shape asserts (demo_celeba.py:60-93, demo_mnist.py:58-71): SAME/stride-s
    Conv2DTranspose multiplies H, W by s; and the independent torch construction agrees."""
    rng = np.random.default_rng(2)
    x = rng.normal(size=(2, h, h, 3))
    w = rng.normal(size=(5, 5, 4, 3))                # [kh,kw,c_out,c_in]
    y = O.conv2d_transpose_fwd(x, w, s)
    assert y.shape == (2, h * s, h * s, 4)
    np.testing.assert_allclose(y, T.conv2d_transpose(torch.from_numpy(x), torch.from_numpy(w), s).numpy(), atol=1e-11)
    # it is the exact adjoint of the SAME conv with the same kernel
    z = rng.normal(size=y.shape)
    assert abs((y * z).sum() - (x * O.conv2d_fwd(z, w, s)).sum()) < 1e-9


def test_reference_generator_shape_asserts():
    for arch, exp in (("mnist", (28, 28, 1)), ("celeba128", (128, 128, 3)), ("celeba64", (64, 64, 3))):
        shapes = M.infer_shapes(M.generator_spec(arch), (M.LATENT[arch],))
        assert shapes[-1] == exp
    sh = M.infer_shapes(M.generator_spec("celeba128"), (100,))
    conv_out = [s for L, s in zip(M.generator_spec("celeba128"), sh) if L["type"] in ("convT", "conv")]
    assert conv_out == [(4, 4, 512), (8, 8, 256), (16, 16, 128), (32, 32, 64), (64, 64, 32), (128, 128, 16), (128, 128, 3)]
    dsh = M.infer_shapes(M.discriminator_spec("celeba128"), (128, 128, 3))
    assert dsh[-2] == (2048,) and dsh[-1] == (1,)
    dsh = M.infer_shapes(M.discriminator_spec("celeba64"), (64, 64, 3))
    assert dsh[-2] == (2048,)
    dsh = M.infer_shapes(M.discriminator_spec("mnist"), (28, 28, 1))
    assert dsh[-2] == (6272,)


def test_param_counts_match_survey():
    cnt = lambda ps: sum(v.size for p in ps for k, v in p.items() if k == "kernel")
    rng = np.random.default_rng(0)
    assert cnt(M.init_params(M.generator_spec("celeba128"), (100,), rng)) == 11738800
    assert cnt(M.init_params(M.discriminator_spec("celeba128"), (128, 128, 3), rng)) == 4368048
    assert cnt(M.init_params(M.generator_spec("celeba64"), (100,), rng)) == 11727200
    assert cnt(M.init_params(M.discriminator_spec("celeba64"), (64, 64, 3), rng)) == 4356448
    assert cnt(M.init_params(M.generator_spec("mnist"), (100,), rng)) == 2280000
    assert cnt(M.init_params(M.discriminator_spec("mnist"), (28, 28, 1), rng)) == 212672


# ---------------------------------------------------------------- BN / Adam / schedule
def test_bn_train_matches_autograd_and_moving_rule():
    rng = np.random.default_rng(3)
    for shape in ((6, 5), (3, 4, 4, 5)):
        x = rng.normal(size=shape) * 2 + 1
        g, b = rng.normal(size=5), rng.normal(size=5)
        mm, mv = np.zeros(5), np.ones(5)
        y, cache, nm, nv = O.bn_train_fwd(x, g, b, mm, mv)
        ax = tuple(range(x.ndim - 1))
        n = x.size // 5
        np.testing.assert_allclose(nm, 0.01 * x.mean(ax))
        var = x.var(ax)
        np.testing.assert_allclose(nv, 0.99 + 0.01 * (var * n / (n - 1) if x.ndim == 4 else var))
        dy = rng.normal(size=shape)
        dx, dg, db = O.bn_train_bwd(dy, g, cache)
        xt = torch.from_numpy(x).requires_grad_(True)
        gt = torch.from_numpy(g).requires_grad_(True)
        bt = torch.from_numpy(b).requires_grad_(True)
        mean = xt.mean(ax)
        yt = gt * (xt - mean) / torch.sqrt(((xt - mean) ** 2).mean(ax) + 1e-3) + bt
        np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-12)
        (yt * torch.from_numpy(dy)).sum().backward()
        np.testing.assert_allclose(dx, xt.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(dg, gt.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(db, bt.grad.numpy(), atol=1e-11)


def test_adam_closed_form_constant_gradient():
    """For a constant gradient g: m_t = g(1-b1^t), v_t = g^2(1-b2^t) so every step moves by
    lr * g/(|g| + eps*sqrt-ish) ~= lr*sign(g) (Keras epsilon-hat form)."""
    th, m, v = np.array([1.0, -2.0]), np.zeros(2), np.zeros(2)
    g = np.array([0.5, -3.0])
    for t in range(1, 6):
        prev = th.copy()
        th, m, v = O.adam_update(th, m, v, g, t, 1e-3)
        np.testing.assert_allclose(m, g * (1 - 0.9 ** t), rtol=1e-12)
        np.testing.assert_allclose(v, g * g * (1 - 0.999 ** t), rtol=1e-12)
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        np.testing.assert_allclose(prev - th, lr_t * m / (np.sqrt(v) + 1e-7), rtol=1e-12)


def test_exponential_decay_q5():
    # callbacks.py:51-61: value = max * 0.96 ** (n_batches / (total/10))
    assert abs(O.exponential_decay(5.0, 0, 20000, 0.96) - 5.0) < 1e-6
    assert abs(O.exponential_decay(5.0, 20000, 20000, 0.96) - 4.8) < 1e-5
    v = O.exponential_decay(5.0, 7914, 202599 / 10, 0.96)     # ~10 epochs of CelebA at B=256
    assert 4.9 < v < 5.0


# ---------------------------------------------------------------- GP closed forms and full step vs autograd
def test_gp_linear_critic_closed_form():
    """For D(x) = <w, x> (+b): GP = (||B w|| - 1)^2 exactly."""
    rng = np.random.default_rng(4)
    st = S.new_state("tiny", rng, np.float64, std=1.0)
    st["dspec"] = [dict(type="flatten"), dict(type="dense", units=1, use_bias=True)]
    w = rng.normal(size=(8 * 8 * 3, 1)) * 0.1
    st["d"] = [dict(), dict(kernel=w, bias=np.zeros(1))]
    r, f = rng.normal(size=(4, 8, 8, 3)), rng.normal(size=(4, 8, 8, 3))
    gp, grads, n = S.gradient_penalty(st, r, f, rng.uniform(size=4))
    bw = O.blur_images(w.reshape(1, 8, 8, 3), 1.0)
    assert abs(gp - (np.linalg.norm(bw) - 1) ** 2) < 1e-12
    # d/dw (||Bw||-1)^2 = 2(||Bw||-1)/||Bw|| * B B w
    nb = np.linalg.norm(bw)
    dw = 2 * (nb - 1) / nb * O.blur_images(bw, 1.0)
    np.testing.assert_allclose(grads[1]["kernel"].reshape(-1), dw.reshape(-1), atol=1e-12)


def _to_t(rnd):
    out = {}
    for k, v in rnd.items():
        out[k] = [torch.from_numpy(m) for m in v] if isinstance(v, list) else torch.from_numpy(v)
    return out


@pytest.mark.parametrize("arch,B,std", [("tiny", 4, 0.05), ("tiny", 3, 1.2), ("tiny_mnist", 4, 0.7)])
def test_step_gradients_match_torch_autograd_fp64(arch, B, std):
    """Explicit formulas (incl. the GP second-order closed form, Q1 vector loss, Q4 inference-BN in the
    D-step) == torch autograd with create_graph double backward, in float64."""
    rng = np.random.default_rng(5)
    st = S.new_state(arch, rng, np.float64, std=std)
    # make biases / BN params non-trivial so every path carries signal
    for ps in (st["g"], st["d"]):
        for p in ps:
            for k in p:
                if k in ("bias", "beta", "moving_mean"):
                    p[k] = rng.normal(size=p[k].shape) * 0.1
                if k in ("gamma", "moving_var"):
                    p[k] = 1 + 0.2 * rng.uniform(size=p[k].shape)
    H, W, C = M.image_shape(arch)
    reals = rng.uniform(-1, 1, size=(B, H, W, C))
    rnd = S.draw_randomness(arch, B, rng, np.float64)
    hp = dict(S.DEFAULT_HP, global_batch_size=B + 1)
    dg, met, fakes = S.discriminator_grads(st, reals, rnd, hp)
    tst = dict(gspec=st["gspec"], dspec=st["dspec"], g=T.to_torch(st["g"], torch.float64),
               d=T.to_torch(st["d"], torch.float64), std=std)
    trnd = _to_t(rnd)
    tdg, tmet, tf = T.discriminator_step_grads(tst, torch.from_numpy(reals), trnd, hp)
    np.testing.assert_allclose(fakes, tf.numpy(), atol=1e-12)
    for k in met:
        assert abs(met[k] - tmet[k]) < 1e-10, k
    for i, g in enumerate(dg):
        for k in g:
            np.testing.assert_allclose(g[k], tdg[i][k].numpy(), atol=1e-9, rtol=1e-9, err_msg=f"D layer {i} {k}")
    gg, upd, gm = S.generator_grads(st, rnd, hp, B)
    tgg, tupd, tgm = T.generator_step_grads(tst, trnd, hp)
    assert abs(gm["gen_loss"] - tgm["gen_loss"]) < 1e-12
    for i, g in enumerate(gg):
        for k in g:
            np.testing.assert_allclose(g[k], tgg[i][k].numpy(), atol=1e-9, rtol=1e-8, err_msg=f"G layer {i} {k}")
    for i in upd:
        np.testing.assert_allclose(upd[i][0], tupd[i][0].numpy(), atol=1e-12)
        np.testing.assert_allclose(upd[i][1], tupd[i][1].numpy(), atol=1e-12)


def test_gp_finite_difference():
    """Independent of any autodiff: central differences of GP w.r.t. a few critic weights."""
    rng = np.random.default_rng(6)
    st = S.new_state("tiny", rng, np.float64, std=0.8)
    r, f = rng.uniform(-1, 1, size=(3, 8, 8, 3)), rng.uniform(-1, 1, size=(3, 8, 8, 3))
    a = rng.uniform(size=3)
    gp, grads, _ = S.gradient_penalty(st, r, f, a)
    for li, key, idx in [(0, "kernel", (2, 3, 1, 4)), (3, "kernel", (0, 4, 5, 7)), (7, "kernel", (17, 0))]:
        w = st["d"][li][key]
        old = w[idx]
        eps = 1e-6
        w[idx] = old + eps
        gp_p = S.gradient_penalty(st, r, f, a, want_grads=False)[0]
        w[idx] = old - eps
        gp_m = S.gradient_penalty(st, r, f, a, want_grads=False)[0]
        w[idx] = old
        fd = (gp_p - gp_m) / (2 * eps)
        assert abs(fd - grads[li][key][idx]) < 1e-6 * max(1, abs(fd)), (li, fd, grads[li][key][idx])


def test_full_train_on_batch_runs_and_counts():
    rng = np.random.default_rng(7)
    st = S.new_state("tiny", rng, np.float32)
    reals = rng.uniform(-1, 1, size=(4, 8, 8, 3)).astype(np.float32)
    rnd = S.draw_randomness("tiny", 4, rng)
    st, met, aux = S.train_on_batch(st, reals, rnd)
    assert st["n_img"] == 4 and st["n_batches"] == 1 and st["d_t"] == 1 and st["g_t"] == 1
    assert set(met) == {"fake_scores", "real_scores", "disc_loss", "gp_term", "norm_term", "gen_loss", "std"}
    assert all(np.isfinite(v) for v in met.values())


def test_input_pipeline_resize_matches_torch_half_pixel_bilinear():
    """N4: oracle normalise + bilinear resize == torch interpolate(align_corners=False) (same half-pixel convention)."""
    import torch.nn.functional as F
    rng = np.random.default_rng(8)
    img = rng.integers(0, 256, size=(2, 21, 17, 3), dtype=np.uint8)
    for out in ((12, 12), (21, 17), (40, 33), (7, 30)):
        got = O.normalize_resize_bilinear(img, out)
        x = (torch.from_numpy(img.astype(np.float64)) - 127.5) / 127.5
        ref = F.interpolate(x.permute(0, 3, 1, 2), size=out, mode="bilinear", align_corners=False).permute(0, 2, 3, 1).numpy()
        np.testing.assert_allclose(got, ref, atol=1e-12)
    same = O.normalize_resize_bilinear(img, (21, 17))
    np.testing.assert_allclose(same, (img.astype(np.float64) - 127.5) / 127.5, atol=1e-14)      # identity size: normalise only


def test_same_padding_rule_agrees_with_an_independent_port_of_keras():
    """Corroboration (not a pin) of one [TF-knowledge] item.  Keras' `imagenet_utils.correct_pad` -- the padding Keras itself puts
    in front of a 'valid' stride-2 conv to reproduce TF's SAME geometry -- survives, ported to PyTorch, in the installed
    `transformers` package (EfficientNet).  For even inputs and stride 2 it must equal the oracle's SAME rule (before = total // 2:
    1 / 2 for k = 5, 0 / 1 for k = 3), for stride 1 the symmetric k // 2."""
    try:
        from transformers.models.efficientnet.modeling_efficientnet import correct_pad
    except Exception as e:                                     # pragma: no cover
        pytest.skip(f"transformers' EfficientNet port not importable: {e}")
    for k in (3, 5, 7):
        left, right, top, bottom = correct_pad(k, adjust=True)
        for n in (8, 28, 64, 128):                             # even inputs, as every feature map of the reference models
            out, before, after = O.same_pads(n, k, 2)
            assert out == n // 2 and (before, after) == (left, right) == (top, bottom), (k, n, before, after, left, right)
        left, right, top, bottom = correct_pad(k, adjust=False)
        out, before, after = O.same_pads(64, k, 1)
        assert out == 64 and (before, after) == (left, right) == (top, bottom)
