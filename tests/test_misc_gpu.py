"""GPU parity: Dense GEMM, column sums, BatchNorm(+LeakyReLU) fwd/bwd, pointwise GP helpers, losses, Adam, RNG."""
import math

import numpy as np
import pytest
import torch

from oracle import np_ops as O
from helpers import dev

pytestmark = pytest.mark.gpu


def test_gemm_variants():
    from blurred_gan_amd import ops
    rng = np.random.default_rng(0)
    for (M, N, K, tA, tB) in [(7, 33, 10, False, False), (10, 33, 7, True, False), (5, 12, 9, False, True), (64, 1, 2048, False, False),
                              (2048, 1, 16, True, False), (256, 8192, 100, False, False), (100, 8192, 256, True, False), (70, 130, 37, False, True),
                              (33, 64, 9, True, True)]:
        A = rng.normal(size=(K, M) if tA else (M, K))
        Bm = rng.normal(size=(N, K) if tB else (K, N))
        bias = rng.normal(size=N)
        ref = (A.T if tA else A) @ (Bm.T if tB else Bm)
        C0 = rng.normal(size=(M, N))
        C = dev(C0)
        ops.gemm(dev(A), dev(Bm), C, M, N, K, tA, tB, bias=dev(bias), beta=0.5, scale=2.0)
        np.testing.assert_allclose(C.cpu().numpy(), 2 * ref + bias + 0.5 * C0, rtol=1e-4, atol=1e-4 * math.sqrt(K))


def test_colsum():
    from blurred_gan_amd import ops
    rng = np.random.default_rng(1)
    for M, N in [(5, 3), (1000, 32), (4096, 70), (3, 512), (100000, 16)]:
        x = rng.normal(size=(M, N))
        ws = torch.empty(ops.colsum_workspace_bytes(M, N) // 4 + 4, device="cuda")
        out = dev(np.ones(N))
        ops.colsum(dev(x), out, M, N, ws, beta=2.0, scale=0.5)
        np.testing.assert_allclose(out.cpu().numpy(), 2.0 + 0.5 * x.sum(0), rtol=1e-4, atol=1e-3)
        ops.colsum(dev(x), out, M, N, ws, square=True)
        np.testing.assert_allclose(out.cpu().numpy(), (x * x).sum(0), rtol=1e-4, atol=1e-3)


@pytest.mark.parametrize("shape", [(16, 40), (3, 4, 4, 16), (64, 8, 8, 32), (256, 70)])
def test_batchnorm_lrelu_fwd_bwd(shape):
    from blurred_gan_amd import ops
    rng = np.random.default_rng(2)
    C = shape[-1]
    M = int(np.prod(shape)) // C
    x = rng.normal(size=shape) * 2 + 0.5
    gamma, beta = 1 + 0.3 * rng.normal(size=C), 0.2 * rng.normal(size=C)
    mm, mv = rng.normal(size=C), 1 + rng.uniform(size=C)
    u, cache, nm, nv = O.bn_train_fwd(x, gamma, beta, mm, mv)
    ref = O.lrelu_fwd(u)
    ws = torch.empty(ops._lib.load().bg_bn_workspace_bytes(M, C) // 4 + 4, device="cuda")
    y, mmd, mvd = torch.empty(shape, device="cuda"), dev(mm), dev(mv)
    sm, si = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_train_fwd(dev(x), y, M, C, dev(gamma), dev(beta), mmd, mvd, sm, si, ws, unbiased=(len(shape) == 4))
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=1e-4, atol=2e-5)
    np.testing.assert_allclose(mmd.cpu().numpy(), nm, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(mvd.cpu().numpy(), nv, rtol=1e-4, atol=1e-5)
    # inference mode
    yi = ops.bn_infer_fwd(dev(x), torch.empty(shape, device="cuda"), M, C, dev(gamma), dev(beta), dev(mm), dev(mv))
    np.testing.assert_allclose(yi.cpu().numpy(), O.lrelu_fwd(O.bn_infer_fwd(x, gamma, beta, mm, mv)), rtol=1e-4, atol=2e-5)
    # backward through lrelu + BN
    dy = rng.normal(size=shape)
    dz = dy * O.lrelu_mask(u)
    dxr, dgr, dbr = O.bn_train_bwd(dz, gamma, cache)
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    dx = ops.bn_train_bwd(dev(dy), y, dev(x), torch.empty(shape, device="cuda"), M, C, dev(gamma), sm, si, dg, db, ws)
    np.testing.assert_allclose(dg.cpu().numpy(), dgr, rtol=2e-4, atol=2e-4 * math.sqrt(M))
    np.testing.assert_allclose(db.cpu().numpy(), dbr, rtol=2e-4, atol=2e-4 * math.sqrt(M))
    np.testing.assert_allclose(dx.cpu().numpy(), dxr, rtol=2e-4, atol=2e-5 * max(1, np.abs(dxr).max()))


def test_pointwise_and_losses():
    from blurred_gan_amd import ops
    rng = np.random.default_rng(3)
    B, n = 6, 8 * 8 * 3
    r, f, a = rng.normal(size=(B, n)), rng.normal(size=(B, n)), rng.uniform(size=B)
    xh = ops.lerp(dev(r), dev(f), dev(a), torch.empty(B, n, device="cuda"))
    np.testing.assert_allclose(xh.cpu().numpy(), r + a[:, None] * (f - r), rtol=1e-5, atol=1e-6)
    nr = ops.row_norm(dev(f), torch.empty(B, device="cuda"))
    np.testing.assert_allclose(nr.cpu().numpy(), np.linalg.norm(f, axis=1), rtol=1e-5)
    gs = ops.gp_seed(dev(f), nr, 0.7, torch.empty(B, n, device="cuda"))
    nn = np.linalg.norm(f, axis=1)
    np.testing.assert_allclose(gs.cpu().numpy(), 0.7 * ((nn - 1) / nn)[:, None] * f, rtol=1e-4, atol=1e-6)
    keep = (rng.uniform(size=(B, n)) > 0.3).astype(np.uint8)
    mg = ops.mul_grad(dev(r), dev(f), torch.empty(B, n, device="cuda"), keep=dev(keep, torch.uint8), alpha=0.3, scale=1 / 0.7)
    np.testing.assert_allclose(mg.cpu().numpy(), r * O.lrelu_mask(f) * keep / 0.7, rtol=1e-5, atol=1e-6)
    y = np.tanh(r)
    tb = ops.tanh_bwd(dev(f), dev(y), torch.empty(B, n, device="cuda"))
    np.testing.assert_allclose(tb.cpu().numpy(), f * (1 - y * y), rtol=1e-5, atol=1e-6)
    ot = ops.outer(dev(a), dev(r[0]), torch.empty(B, n, device="cuda"))
    np.testing.assert_allclose(ot.cpu().numpy(), np.outer(a, r[0]), rtol=1e-6)
    # D loss: wgan.py:130, 277-285 incl. the [B]-vector quirk scale
    fs, rs, norms = rng.normal(size=B), rng.normal(size=B), 1 + rng.uniform(size=B)
    fs[0] = 0.0
    dfs, drs, met = torch.empty(B, device="cuda"), torch.empty(B, device="cuda"), torch.empty(8, device="cuda")
    ops.wgangp_d_loss(dev(fs), dev(rs), dev(norms), 1 / 32, 10.0, 1e-4, float(B), dfs, drs, met)
    gp = ((norms - 1) ** 2).mean()
    norm_term = 1e-4 * (np.abs(fs) + np.abs(rs))
    m = met.cpu().numpy()
    np.testing.assert_allclose(m[:6], [fs.mean(), rs.mean(), (fs - rs).sum() / 32 + 10 * gp + norm_term.mean(), 10 * gp,
                                       norm_term.mean(), gp], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(dfs.cpu().numpy(), B / 32 + 1e-4 * np.sign(fs), rtol=1e-6)
    np.testing.assert_allclose(drs.cpu().numpy(), -B / 32 + 1e-4 * np.sign(rs), rtol=1e-6)
    ds, gm = torch.empty(B, device="cuda"), torch.empty(4, device="cuda")
    ops.wgan_g_loss(dev(fs), 1 / 32, ds, gm)
    np.testing.assert_allclose(gm.cpu().numpy()[:2], [fs.mean(), -fs.sum() / 32], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(ds.cpu().numpy(), -1 / 32)


def test_adam_matches_oracle():
    from blurred_gan_amd import ops
    rng = np.random.default_rng(4)
    n = 10007
    th, m, v = rng.normal(size=n), np.zeros(n), np.zeros(n)
    thd, md, vd = dev(th), dev(m), dev(v)
    for t in range(1, 4):
        g = rng.normal(size=n) * 10.0 ** rng.integers(-6, 1, size=n)
        th, m, v = O.adam_update(th, m, v, g, t, 1e-3)
        lr_t = 1e-3 * math.sqrt(1 - 0.999 ** t) / (1 - 0.9 ** t)
        ops.adam(thd, md, vd, dev(g), lr_t)
    np.testing.assert_allclose(thd.cpu().numpy(), th, rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(md.cpu().numpy(), m, rtol=1e-5, atol=2e-7)    # fp32 cancellation between steps (|g| up to ~3)
    np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=3e-5, atol=1e-12)   # (1 - 0.999f) differs from 1e-3 by 1.3e-5 in float32, as in TF


def test_rng_statistics_and_determinism():
    from blurred_gan_amd import ops
    n = 1 << 20
    u = ops.uniform(torch.empty(n, device="cuda"), 123, 0).cpu().numpy()
    assert u.min() >= 0.0 and u.max() < 1.0
    assert abs(u.mean() - 0.5) < 2e-3 and abs(u.var() - 1 / 12) < 2e-3
    u2 = ops.uniform(torch.empty(n, device="cuda"), 123, 0).cpu().numpy()
    assert np.array_equal(u, u2)                                        # counter-based: same (seed, offset) -> same stream
    u3 = ops.uniform(torch.empty(n, device="cuda"), 123, n // 4).cpu().numpy()
    assert not np.array_equal(u, u3) and abs(np.corrcoef(u, u3)[0, 1]) < 5e-3
    k = ops.keep_mask(torch.empty(n + 3, dtype=torch.uint8, device="cuda"), 0.7, 5, 0).cpu().numpy()
    assert set(np.unique(k)) <= {0, 1} and abs(k.mean() - 0.7) < 2e-3


def _philox4x32_10(ctr, key, c2):
    """Philox4x32-10 (Salmon et al., SC'11) on arrays of 64-bit counters: counter words (lo, hi, c2, 0), key (lo, hi)."""
    c = [(ctr & 0xFFFFFFFF).astype(np.uint64), (ctr >> np.uint64(32)).astype(np.uint64), np.full_like(ctr, c2), np.zeros_like(ctr)]
    k0, k1 = np.uint64(key & 0xFFFFFFFF), np.uint64(key >> 32)
    M0, M1, mask = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57), np.uint64(0xFFFFFFFF)
    for _ in range(10):
        p0, p1 = M0 * c[0], M1 * c[2]
        c = [(p1 >> np.uint64(32)) ^ c[1] ^ k0, p1 & mask, (p0 >> np.uint64(32)) ^ c[3] ^ k1, p0 & mask]
        k0, k1 = (k0 + np.uint64(0x9E3779B9)) & mask, (k1 + np.uint64(0xBB67AE85)) & mask
    return np.stack(c, axis=1)


@pytest.mark.parametrize("n,shift", [(4096, 0), (1000, 0), (1003, 1), (37, 5), (16 * 700 + 9, 16)])
def test_rng_streams_are_counter_addressed(n, shift):
    """The layout the step programs and the data-parallel offsets rely on: element e of a draw is word e % 4 of the Philox block
    with counter offset + e / 4 (third counter word 0 for uniforms, 1 for keep masks), whatever the output's alignment and
    however many elements a thread forms."""
    from blurred_gan_amd import ops
    seed, offset = 0x1234567, 77
    blocks = _philox4x32_10(np.arange(offset, offset + (n + 3) // 4, dtype=np.uint64), seed, 1).reshape(-1)[:n]
    u = (blocks >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0)
    buf = torch.zeros(n + shift + 16, dtype=torch.uint8, device="cuda")
    k = ops.keep_mask(buf[shift:shift + n], 0.7, seed, offset).cpu().numpy()
    assert np.array_equal(k, (u < np.float32(0.7)).astype(np.uint8))
    assert buf[:shift].sum().item() == 0 and buf[shift + n:].sum().item() == 0          # nothing written outside
    blocks0 = _philox4x32_10(np.arange(offset, offset + (n + 3) // 4, dtype=np.uint64), seed, 0).reshape(-1)[:n]
    got = ops.uniform(torch.empty(n, device="cuda"), seed, offset).cpu().numpy()
    assert np.array_equal(got, (blocks0 >> np.uint64(8)).astype(np.float32) * np.float32(1.0 / 16777216.0))


@pytest.mark.parametrize("src_hw,dst_hw,C", [((218, 178), (128, 128), 3), ((28, 28), (28, 28), 1), ((16, 20), (40, 33), 3)])
def test_input_pipeline_kernel(src_hw, dst_hw, C):
    """N4 (demo_celeba.py:22-35): uint8 -> normalise -> bilinear resize on the device vs the oracle."""
    from blurred_gan_amd import ops
    rng = np.random.default_rng(9)
    img = rng.integers(0, 256, size=(3,) + src_hw + (C,), dtype=np.uint8)
    dst = torch.empty((3,) + dst_hw + (C,), device="cuda")
    ops.u8_normalize_resize(torch.from_numpy(img).cuda(), dst)
    ref = O.normalize_resize_bilinear(img, dst_hw)
    np.testing.assert_allclose(dst.cpu().numpy(), ref, rtol=1e-5, atol=2e-6)
    assert dst.min().item() >= -1.0 - 1e-6 and dst.max().item() <= 1.0 + 1e-6


def test_separable_batchnorm_pieces_equal_fused():
    """The SyncBN entry points (stats -> [all-reduce] -> finalize -> apply; bwd stats -> [all-reduce] -> bwd apply) with a
    single replica reproduce the fused bg_bn_train_fwd / bg_bn_train_bwd bit for bit."""
    from blurred_gan_amd import ops
    rng = np.random.default_rng(12)
    shape, C = (32, 8, 8, 64), 64
    M = int(np.prod(shape)) // C
    x, dy = dev(rng.normal(size=shape) * 2 + 0.3), dev(rng.normal(size=shape))
    gamma, beta = dev(1 + 0.2 * rng.normal(size=C)), dev(0.1 * rng.normal(size=C))
    ws = torch.empty(ops._lib.load().bg_bn_workspace_bytes(M, C) // 4 + 4, device="cuda")
    mm1, mv1, mm2, mv2 = (dev(np.zeros(C)), dev(np.ones(C)), dev(np.zeros(C)), dev(np.ones(C)))
    y1, y2 = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
    m1, i1, m2, i2 = (torch.empty(C, device="cuda") for _ in range(4))
    ops.bn_train_fwd(x, y1, M, C, gamma, beta, mm1, mv1, m1, i1, ws)
    sums = torch.empty(2 * C, device="cuda")
    ops.bn_stats(x, M, C, sums, ws)
    ops.bn_finalize(sums, M, C, m2, i2, mm2, mv2)
    ops.bn_apply(x, y2, M, C, gamma, beta, m2, i2)
    for a, b in ((y1, y2), (m1, m2), (i1, i2), (mm1, mm2), (mv1, mv2)):
        assert torch.equal(a, b)
    dx1, dx2 = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
    dg, db = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_train_bwd(dy, y1, x, dx1, M, C, gamma, m1, i1, dg, db, ws)
    ops.bn_bwd_stats(dy, y1, x, M, C, m1, i1, sums, ws)
    ops.bn_bwd_apply(dy, y1, x, dx2, M, M, C, gamma, m1, i1, sums)
    assert torch.equal(dx1, dx2) and torch.equal(db, sums[:C]) and torch.equal(dg, sums[C:])


@pytest.mark.parametrize("shape,alpha", [((32, 8, 8, 64), 0.2), ((16, 4, 4, 64), 0.0), ((5, 3, 3, 7), 0.2), ((64, 16, 16, 128), 0.0)])
def test_batchnorm_backward_without_saved_activation(shape, alpha):
    """y = NULL: the sign of the activation is re-derived from x with the forward's expression -- every output of the fused and
    of the separable backward is bit-identical to the route that reads the saved y (float4 and scalar kernels, ReLU and LeakyReLU;
    values sit ON zero too: beta = 0, x = mean exactly for a constant channel)."""
    from blurred_gan_amd import ops
    rng = np.random.default_rng(5)
    C = shape[-1]
    M = int(np.prod(shape)) // C
    xn = rng.normal(size=shape) * 2 + 0.3
    xn[..., 0] = 1.25                                  # a constant channel: (x - mean) == 0, v == beta
    x, dy = dev(xn), dev(rng.normal(size=shape))
    bn_ = 0.1 * rng.normal(size=C)
    bn_[0] = 0.0
    gamma, beta = dev(1 + 0.2 * rng.normal(size=C)), dev(bn_)
    ws = torch.empty(ops._lib.load().bg_bn_workspace_bytes(M, C) // 4 + 4, device="cuda")
    mm, mv = dev(np.zeros(C)), dev(np.ones(C))
    y = torch.empty(shape, device="cuda")
    m, iv = torch.empty(C, device="cuda"), torch.empty(C, device="cuda")
    ops.bn_train_fwd(x, y, M, C, gamma, beta, mm, mv, m, iv, ws, lrelu_alpha=alpha)
    outs = []
    for yy in (y, None):
        dx, dx2 = torch.empty(shape, device="cuda"), torch.empty(shape, device="cuda")
        dg, db, sums = torch.empty(C, device="cuda"), torch.empty(C, device="cuda"), torch.empty(2 * C, device="cuda")
        ops.bn_train_bwd(dy, yy, x, dx, M, C, gamma, m, iv, dg, db, ws, lrelu_alpha=alpha, beta=beta)
        ops.bn_bwd_stats(dy, yy, x, M, C, m, iv, sums, ws, lrelu_alpha=alpha, gamma=gamma, beta=beta)
        ops.bn_bwd_apply(dy, yy, x, dx2, M, M, C, gamma, m, iv, sums, lrelu_alpha=alpha, beta=beta)
        outs.append((dx, dg, db, sums, dx2))
    for a, b in zip(*outs):
        assert torch.equal(a, b)
    assert torch.equal(outs[1][0], outs[1][4])


def test_bn_fold_many_equals_per_layer_folds():
    """bg_bn_fold_many_f32: several layers' inference folds in one launch, bit-identical to one bg_bn_fold_f32 per layer (views into
    a flat buffer at odd offsets, different channel counts and epsilons)."""
    from blurred_gan_amd import ops
    rng = np.random.default_rng(8)
    Cs, eps = [512, 256, 64, 7, 300], [1e-3, 1e-3, 1e-5, 1e-3, 2e-3]
    flat = dev(rng.normal(size=4 * sum(Cs) + 3))
    layers, want, o = [], [], 1
    for C, e in zip(Cs, eps):
        g, b, mm, mv = (flat[o + k * C:o + (k + 1) * C] for k in range(4))
        mv.abs_()
        o += 4 * C
        out = torch.zeros(2 * C, device="cuda")
        ref = torch.zeros(2 * C, device="cuda")
        ops.bn_fold(g, b, mm, mv, e, ref[:C], ref[C:])
        layers.append((g, b, mm, mv, e, out[:C], out[C:]))
        want.append((ref, out))
    ops.bn_fold_many(layers)
    for ref, out in want:
        assert torch.equal(ref, out) and torch.isfinite(out).all()


def test_gp_seed_zero_norm_quirk_and_guard():
    """A sample with an exactly-zero input gradient: the reference formula (n-1)/n * g is NaN there (as tf.norm's gradient at 0);
    the guarded entry point takes the subgradient 0 for that sample and leaves the others untouched."""
    from blurred_gan_amd import ops
    g = torch.randn(3, 40, device="cuda")
    g[1] = 0.0
    n = ops.row_norm(g, torch.empty(3, device="cuda"))
    ref = ops.gp_seed(g, n, 0.5, torch.empty_like(g)).cpu()
    guarded = ops.gp_seed(g, n, 0.5, torch.empty_like(g), zero_norm_guard=True).cpu()
    assert torch.isnan(ref[1]).all() and torch.isfinite(ref[[0, 2]]).all()
    assert (guarded[1] == 0).all() and torch.equal(guarded[[0, 2]], ref[[0, 2]])
