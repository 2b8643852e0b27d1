"""GPU parity of the whole hot path: BlurredWGANGP.train_on_batch (through the reference-shaped Python API and
the C ABI) vs the numpy oracle, on identical weights and injected randomness."""
import copy

import numpy as np
import pytest
import torch

from oracle import step as S
from helpers import (load_oracle_weights, product_grads, product_slots, oracle_grad_list, oracle_weight_list, check_grad_quality, rel_l2,
                     to_float32_state, to_float32_randomness, product_lrelu_branches, arm_branch_capture, cosine)

pytestmark = pytest.mark.gpu


def _make(arch, B, std, seed=0, gbs=None, **kw):
    import blurred_gan_amd as bg
    from blurred_gan_amd import models
    rng = np.random.default_rng(seed)
    st = S.new_state(arch, rng, np.float64, std=std)
    for ps in (st["g"], st["d"]):            # non-trivial biases / BN parameters
        for p in ps:
            for k in p:
                if k in ("bias", "beta", "moving_mean"):
                    p[k] = rng.normal(size=p[k].shape) * 0.1
                if k in ("gamma", "moving_var"):
                    p[k] = 1 + 0.2 * rng.uniform(size=p[k].shape)
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=std, global_batch_size=gbs or B, batch_size=B)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir="/tmp/bg_test_logs"), **kw)
    load_oracle_weights(gen, st["g"])
    load_oracle_weights(disc, st["d"])
    H, W, C = models.IMAGE_SHAPE[arch]
    reals = rng.uniform(-1, 1, size=(B, H, W, C))
    return gan, st, reals, rng


@pytest.mark.parametrize("arch,B,std", [("tiny", 4, 0.05), ("tiny", 5, 1.2), ("tiny_mnist", 4, 0.7), ("mnist", 3, 0.05),
                                        ("celeba64", 2, 1.0), ("celeba128", 2, 2.0),      # the headline architectures, every layer shape of C2 / C4
                                        ("celeba64", 64, 5.0)])     # ... and at a batch that takes the position-major / tap-skipping / split-K paths
def test_gradients_match_oracle(arch, B, std):
    gan, st, reals, rng = _make(arch, B, std, gbs=B + 1)
    rnd = S.draw_randomness(arch, B, rng, np.float64)
    hp = dict(S.DEFAULT_HP, global_batch_size=B + 1)
    dg, met, fakes = S.discriminator_grads(st, reals, rnd, hp)
    # run the product D-step but look at the gradients before they are consumed: lr = 0 keeps weights fixed
    gan.discriminator.optimizer.learning_rate = 0.0
    gan.generator.optimizer.learning_rate = 0.0
    out = gan.train_on_batch(reals.astype(np.float32), randomness=rnd)
    names = gan.metrics_names
    got = dict(zip(names, out))
    # This test lets product and oracle take their OWN LeakyReLU branches.  At the small batches that is harmless (a few
    # thousand units, none near the kink); at batch 64 (12 M critic and 17 M generator units) a unit within float32 rounding of
    # zero may sit on different branches in the two, and that one unit moves a bias-gradient element by up to 0.5 % and the
    # generator's tensors by 1e-3 in relative L2 -- the float32 ORACLE deviates from its float64 run by 8e-4 ... 7e-3 there.
    # So the batch-64 case is held to loose elementwise bounds and to max(5e-3 generator / 2e-3 critic, 3 x the float32 oracle)
    # in relative L2; the SHARP comparison, with the branches shared, is test_gradients_match_oracle_on_the_same_relu_branches
    # (same shapes: 9.7e-6 / 1.1e-6).
    big = B >= 64
    pg = product_grads(gan.discriminator)
    for a, b in zip(pg, oracle_grad_list(dg)):
        scale = max(np.abs(b).max(), 1e-6)
        np.testing.assert_allclose(a, b.reshape(a.shape), rtol=2e-3, atol=(1e-2 if big else 2e-4) * scale)
    gg, upd, gm = S.generator_grads(st, rnd, hp, B)
    pgg = product_grads(gan.generator)
    for a, b in zip(pgg, oracle_grad_list(gg)):
        scale = max(np.abs(b).max(), 1e-6)
        np.testing.assert_allclose(a, b.reshape(a.shape), rtol=2e-3, atol=(1e-2 if big else 2e-4) * scale)
    dg32 = gg32 = None
    if big:
        st32, rnd32 = to_float32_state(st), to_float32_randomness(rnd)
        dg32 = oracle_grad_list(S.discriminator_grads(st32, reals.astype(np.float32), rnd32, hp)[0])
        gg32 = oracle_grad_list(S.generator_grads(st32, rnd32, hp, B)[0])
    check_grad_quality(pg, oracle_grad_list(dg), "d", f"{arch} B={B} critic", dg32, l2_bound=2e-3 if big else None, cos_bound=1e-5 if big else 1e-6)
    check_grad_quality(pgg, oracle_grad_list(gg), "g", f"{arch} B={B} generator", gg32, l2_bound=5e-3 if big else None, cos_bound=1e-5 if big else 1e-6)
    np.testing.assert_allclose(gan.images[0].cpu().numpy(), fakes, rtol=1e-4, atol=1e-5)
    for k in ("real_scores", "disc_loss", "gp_term", "norm_term"):
        assert abs(got[k] - met[k]) < 1e-4 * max(1, abs(met[k])), (k, got[k], met[k])
    assert abs(got["gen_loss"] - gm["gen_loss"]) < 1e-4 * max(1, abs(gm["gen_loss"]))

    assert abs(got["fake_scores"] - 0.5 * (met["fake_scores"] + gm["fake_scores_g"])) < 1e-4      # Q6
    assert abs(got["std"] - std) < 1e-7 and got["loss"] == 0.0


@pytest.mark.parametrize("arch,B,std", [("tiny", 4, 0.9), ("tiny_mnist", 3, 0.05)])
def test_three_training_steps_match_oracle(arch, B, std):
    """Weights, Adam slots, BN moving statistics and counters after 3 full steps (float32 oracle vs HIP)."""
    gan, st64, reals, rng = _make(arch, B, std)
    st = copy.deepcopy(st64)
    for key in ("g", "d", "g_m", "g_v", "d_m", "d_v"):
        st[key] = [{k: v.astype(np.float64) for k, v in p.items()} for p in st[key]]
    hp = dict(S.DEFAULT_HP, global_batch_size=B)
    for it in range(3):
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        r = rng.uniform(-1, 1, size=reals.shape)
        st, met, _ = S.train_on_batch(st, r, rnd, hp)
        out = gan.train_on_batch(r.astype(np.float32), randomness=rnd)
    assert int(gan.n_img) == 3 * B and int(gan.n_batches) == 3
    for model, key in ((gan.generator, "g"), (gan.discriminator, "d")):
        for a, b in zip(model.get_weights(), oracle_weight_list(st[key])):
            np.testing.assert_allclose(a, b.reshape(a.shape), rtol=1e-3, atol=2e-4)


@pytest.mark.parametrize("arch,B,std,steps", [("celeba64", 8, 5.0, 3), ("mnist", 8, 0.05, 3), ("celeba128", 4, 5.0, 2)])
def test_real_architecture_training_steps_match_oracle(arch, B, std, steps):
    """Consecutive full steps with the REAL learning rate at the real layer shapes (wgan.py:140-141,166-167): after every
    train_on_batch the weights, both Adam slots, the BatchNorm moving statistics, the optimiser iteration counts and the
    counters of the HIP path equal the float64 oracle's.  The product runs its steps back to back, untouched: that is what
    exercises the flat-buffer Adam (bias correction at t = 1, 2, 3), the refresh of the transposed weight copies the forward
    kernels read, and the G-step's moving-statistics update at 16 M parameters.

    Two things make the comparison well-posed.  (1) The ORACLE is re-synchronised to the product's state before each step: a
    free-running comparison of weights is ill-conditioned (Adam's first updates are lr * g / (|g| + 1e-7): a weight whose
    gradient is float32 noise steps +lr in one implementation and -lr in the other).  (2) The oracle's backward is given the
    LeakyReLU branch decisions the product took in that step (oracle.step `force`): the gradient is discontinuous where a
    pre-activation crosses zero, about one of the 2 M units of a generator pass sits within float32 rounding of the kink, and
    that one unit moved a whole tensor's relative L2 error to 2e-3 before the branches were shared (round 3: measured
    2.0e-3 on the generator's Dense kernel at celeba64 / batch 8 / step 2, 9.7e-6 with the branches shared --
    test_gradients_match_oracle_on_the_same_relu_branches).  With both in place the slots agree to a relative L2 error of 1e-4."""
    from helpers import sync_oracle_from_product
    gan, st, reals, rng = _make(arch, B, std, seed=5)
    hp = dict(S.DEFAULT_HP, global_batch_size=B)
    lr = hp["learning_rate"]
    worst = {}
    for it in range(steps):
        if it:
            sync_oracle_from_product(st, gan)               # the product itself is NOT reloaded
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        r = rng.uniform(-1, 1, size=reals.shape)
        arm_branch_capture(gan)
        got = dict(zip(gan.metrics_names, gan.train_on_batch(r.astype(np.float32), randomness=rnd)))
        st, met, _ = S.train_on_batch(st, r, rnd, hp, force=product_lrelu_branches(gan, B))
        for k in ("disc_loss", "gen_loss", "gp_term", "real_scores", "fake_scores"):
            assert abs(got[k] - met[k]) < 1e-3 * (abs(met[k]) + 0.1), (it, k, got[k], met[k])
        assert int(gan.n_img) == (it + 1) * B == st["n_img"] and int(gan.n_batches) == it + 1 == st["n_batches"]
        assert gan.generator.optimizer.iterations == st["g_t"] == it + 1
        assert gan.discriminator.optimizer.iterations == st["d_t"] == it + 1
        for model, key in ((gan.generator, "g"), (gan.discriminator, "d")):
            names = [n for (l, n, _, _, tr) in model.store.entries if id(l) in {id(x) for x in model._own_layers()}]
            for a, b, name in zip(model.get_weights(), oracle_weight_list(st[key]), names):     # weights AND BN moving statistics
                b = b.reshape(a.shape)
                err = np.abs(a - b)
                if name.startswith("moving"):
                    np.testing.assert_allclose(a, b, rtol=2e-4, atol=2e-6, err_msg=f"{key} {name} step {it}")
                    continue
                # one Adam step moves a weight by at most ~lr; where the gradient is float32 noise the SIGN of that move is too
                assert err.max() <= 2.0 * lr + 1e-6, (key, name, a.shape, err.max())
                bad = float((err > 1e-3 * np.abs(b) + 1e-4).mean())
                worst[key + "_w_bad_frac"] = max(worst.get(key + "_w_bad_frac", 0.0), bad)
                assert bad <= 0.02, (key, name, a.shape, bad)
            for slot in ("m", "v"):
                for a, b in zip(product_slots(model, slot), oracle_grad_list(st[f"{key}_{slot}"])):
                    b = np.asarray(b, dtype=np.float64).reshape(a.shape)
                    e = float(np.abs(a - b).max() / max(np.abs(b).max(), 1e-30))
                    worst[key + "_" + slot] = max(worst.get(key + "_" + slot, 0.0), e)
                    assert e <= (1e-2 if key == "g" else 3e-3), (key, slot, a.shape, e, it)
                    if a.size > 1:          # (the critic's Dense bias is a scalar residue of +-B/gbs terms: 1.7e-4 relative by itself)
                        l2 = rel_l2(a, b)
                        worst[key + "_" + slot + "_l2"] = max(worst.get(key + "_" + slot + "_l2", 0.0), l2)
                        assert l2 <= 1e-4, (key, slot, a.shape, l2, it)
    print(f"{arch} B={B} {steps} steps: worst deviations {worst}")


@pytest.mark.parametrize("arch,B,std,steps,seed", [("celeba64", 8, 5.0, 2, 5), ("mnist", 8, 0.05, 2, 5), ("celeba64", 64, 5.0, 1, 0),
                                                   ("celeba128", 4, 5.0, 2, 5)])
def test_gradients_match_oracle_on_the_same_relu_branches(arch, B, std, steps, seed):
    """The sharp form of the gradient comparison.  The step's gradient is a DISCONTINUOUS function of the weights: every
    LeakyReLU unit whose pre-activation crosses zero switches its derivative between 1 and 0.3.  A 64x64 generator pass at
    batch 8 has 2.1 M such units; about one per step lands within float32 rounding of the kink, where the HIP path and the
    float64 oracle legitimately sit on different branches -- and that ONE unit moves the BatchNorm backward sums of its
    channel coherently, which is what the per-tensor deviations of 1e-3 in test_gradients_match_oracle are (signature: the
    error appears at one layer's d(beta), 20x smaller in its d(gamma), and rides down the stack from there; the layers above
    it agree to 1e-5: tests/step_error.py).  Here the oracle's backward is given the branch decisions the product actually
    took (oracle.step `force`): both then differentiate the same piecewise-linear function, and every gradient of both
    networks agrees to a relative L2 error of 5e-5 (generator; measured <= 9.7e-6) / 1e-5 (critic; measured <= 1.1e-6) with
    cosine 1 - 1e-9 -- on consecutive steps with the real learning rate,
    the oracle re-synchronised to the product's state before each (the gradients are read back from Adam's first moment)."""
    from helpers import sync_oracle_from_product
    gan, st, reals, rng = _make(arch, B, std, seed=seed)
    hp = dict(S.DEFAULT_HP, global_batch_size=B)
    for mod in (gan.generator, gan.discriminator):
        mod.store.ensure_opt_state()
    worst = {"g": (0.0, 0.0), "d": (0.0, 0.0)}
    for it in range(steps):
        if it:
            sync_oracle_from_product(st, gan)
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        r = rng.uniform(-1, 1, size=reals.shape)
        m_old = {k: [a.astype(np.float64) for a in product_slots(mod, "m")] for k, mod in (("g", gan.generator), ("d", gan.discriminator))}
        arm_branch_capture(gan)
        gan.train_on_batch(r.astype(np.float32), randomness=rnd)
        st, met, aux = S.train_on_batch(st, r, rnd, hp, force=product_lrelu_branches(gan, B))
        for key, mod in (("d", gan.discriminator), ("g", gan.generator)):
            prod = [(a.astype(np.float64) - float(np.float32(0.9)) * b) / (1.0 - float(np.float32(0.9)))
                    for a, b in zip(product_slots(mod, "m"), m_old[key])]
            for i, (a, b) in enumerate(zip(prod, oracle_grad_list(aux[f"{key}_grads"]))):
                b = np.asarray(b, np.float64).reshape(a.shape)
                if a.size == 1 or not np.any(b):          # the critic's Dense bias: a 1e-4-sized residue of +-B/gbs terms
                    assert abs(float(a.ravel()[0] - b.ravel()[0])) <= 2e-3 * abs(float(b.ravel()[0])) + 1e-6, (key, i, it)
                    continue
                l2, c = rel_l2(a, b), 1.0 - cosine(a, b)
                worst[key] = (max(worst[key][0], l2), max(worst[key][1], c))
                assert l2 <= (5e-5 if key == "g" else 1e-5) and c <= 1e-9, (arch, key, i, a.shape, it, l2, c)
    print(f"[same branches] {arch} B={B} {steps} step(s): worst rel-L2 / (1-cos): generator {worst['g'][0]:.1e} / {worst['g'][1]:.0e}, "
          f"critic {worst['d'][0]:.1e} / {worst['d'][1]:.0e}")


def _full_batch_step_matches_oracle(arch, B, std):
    """One BASELINE.json configuration at its REAL batch: one full train_on_batch against the
    float64 numpy oracle on identical injected randomness (learning rate 0 so the gradients can be read back), the oracle
    differentiating on the LeakyReLU branches the product took (48 M critic units and 67 M generator units per step: a few dozen
    sit within float32 rounding of the kink, and each one that lands on the other branch moves a bias-gradient element -- a sum
    of 131 k terms of both signs -- by up to 0.5 %; see test_gradients_match_oracle_on_the_same_relu_branches).  The oracle
    needs about half a minute of host time at this batch."""
    gan, st, reals, rng = _make(arch, B, std, seed=9)
    rnd = S.draw_randomness(arch, B, rng, np.float64)
    hp = dict(S.DEFAULT_HP, global_batch_size=B)
    gan.discriminator.optimizer.learning_rate = 0.0
    gan.generator.optimizer.learning_rate = 0.0
    arm_branch_capture(gan)
    got = dict(zip(gan.metrics_names, gan.train_on_batch(reals.astype(np.float32), randomness=rnd)))
    force = product_lrelu_branches(gan, B)
    dg, met, fakes = S.discriminator_grads(st, reals, rnd, hp, force)
    gg, upd, gm = S.generator_grads(st, rnd, hp, B, force)
    worst = {}
    for key, prod, ora in (("d", product_grads(gan.discriminator), oracle_grad_list(dg)), ("g", product_grads(gan.generator), oracle_grad_list(gg))):
        for i, (a, b) in enumerate(zip(prod, ora)):
            b = np.asarray(b, np.float64).reshape(a.shape)
            # elementwise: each filter-gradient element sums up to 5e5 float32 products; 1e-4 of the variable's largest entry
            np.testing.assert_allclose(a, b, rtol=2e-3, atol=1e-4 * max(np.abs(b).max(), 1e-6), err_msg=f"{key}{i:02d}")
            if a.size > 1:
                l2, c = rel_l2(a, b), 1.0 - cosine(a, b)
                worst[key] = (max(worst.get(key, (0, 0))[0], l2), max(worst.get(key, (0, 0))[1], c))
                assert l2 <= 2e-5 and c <= 1e-9, (key, i, a.shape, l2, c)          # measured: generator 3.2e-6, critic 3.7e-6
    print(f"[same branches] {arch} B={B}: worst rel-L2 / (1-cos): generator {worst['g'][0]:.1e} / {worst['g'][1]:.0e}, "
          f"critic {worst['d'][0]:.1e} / {worst['d'][1]:.0e}")
    np.testing.assert_allclose(gan.images[0].cpu().numpy(), fakes, rtol=1e-4, atol=1e-5)
    for k in ("real_scores", "disc_loss", "gp_term", "norm_term"):
        assert abs(got[k] - met[k]) < 1e-4 * max(1, abs(met[k])), (k, got[k], met[k])
    assert abs(got["gen_loss"] - gm["gen_loss"]) < 1e-4 * max(1, abs(gm["gen_loss"]))

    # ---- the UN-FORCED leg: what sharing the branches could hide is a product that takes the wrong LeakyReLU branch
    # at a unit that is NOT marginal.  So: the float64 oracle on its OWN branches; every unit where the product's branch differs
    # must be one whose float64 pre-activation lies within float32 rounding of the kink, there must be only a handful of them
    # among the 1.2e8 units of the step, and the critic's gradient elements that leave the forced-branch bound must fit what those
    # units can touch (a flipped unit of conv layer l changes one output channel's 25 * Cin_l filter column and its bias element).
    from oracle import models as M, np_ops as O
    own, pre = {}, {}

    def collect(name, spec, cache):
        own[name] = [np.asarray(c["m"]) for L, c in zip(spec, cache) if L["type"] == "lrelu"]
        pre[name] = [np.asarray(c["x"]) for L, c in zip(spec, cache) if L["type"] == "lrelu"]

    _, c = S.critic_fwd(st, fakes, True, rnd["mask_fake"]); collect("fake", st["dspec"], c)
    _, c = S.critic_fwd(st, reals, True, rnd["mask_real"]); collect("real", st["dspec"], c)
    xhat = reals + rnd["alpha"].reshape(B, 1, 1, 1) * (fakes - reals)
    _, c = S.critic_fwd(st, xhat, False); collect("hat", st["dspec"], c)
    fg, c = M.forward(st["gspec"], st["g"], rnd["z_g"], training=True); collect("g", st["gspec"], c)
    ks, se, _ = O.blur_policy(st["std"], fg.shape[1], fg.shape[2])
    _, c = M.forward(st["dspec"], st["d"], O.gaussian_blur(fg, se, ks), training=False); collect("d_gstep", st["dspec"], c)
    keep = {"fake": rnd["mask_fake"], "real": rnd["mask_real"]}
    flips, units, worst_margin, d_budget = {}, 0, 0.0, 0
    conv_cin = []
    shp = M.infer_shapes(st["dspec"], M.image_shape(arch))
    prev_c = M.image_shape(arch)[-1]
    for L, sh in zip(st["dspec"], shp):
        if L["type"] == "conv":
            conv_cin.append(prev_c)
            prev_c = sh[-1]
    for name in own:
        n_flip = 0
        for li, (mo, z, mp_) in enumerate(zip(own[name], pre[name], force[name])):
            diff = np.asarray(mp_, np.float64).reshape(mo.shape) != mo
            if name in keep:                              # a unit Dropout zeroed reads as `alpha` in the product's activations
                diff &= np.asarray(keep[name][li]).reshape(mo.shape).astype(bool)
            units += mo.size
            k = int(diff.sum())
            if k:
                n_flip += k
                rms = float(np.sqrt(np.mean(z * z)))
                worst_margin = max(worst_margin, float(np.abs(z[diff]).max()) / rms)
                if name in ("fake", "real", "hat"):
                    d_budget += k * (25 * conv_cin[li] + 1)
        flips[name] = n_flip
    total = sum(flips.values())
    print(f"[own branches] {arch} B={B}: {total} of {units} LeakyReLU units on another branch than the float64 oracle {flips}; "
          f"largest |pre-activation| among them {worst_margin:.1e} of its layer's rms")
    assert total <= 400, flips                            # measured: 36 (generator 30, critic 6)
    assert worst_margin <= 5e-5, worst_margin             # every one of them within float32 rounding of the kink (measured 6.3e-6 of the layer rms)
    dg2, _, _ = S.discriminator_grads(st, reals, rnd, hp)
    gg2, _, _ = S.generator_grads(st, rnd, hp, B)
    outside = 0
    for i, (a, b) in enumerate(zip(product_grads(gan.discriminator), oracle_grad_list(dg2))):
        b = np.asarray(b, np.float64).reshape(a.shape)
        outside += int((np.abs(a - b) > 2e-3 * np.abs(b) + 1e-4 * max(np.abs(b).max(), 1e-6)).sum())
        if a.size > 1:
            assert rel_l2(a, b) <= 2e-3, ("d", i, rel_l2(a, b))
    worst_g = 0.0
    for i, (a, b) in enumerate(zip(product_grads(gan.generator), oracle_grad_list(gg2))):
        b = np.asarray(b, np.float64).reshape(a.shape)
        if a.size > 1:
            worst_g = max(worst_g, rel_l2(a, b))
    print(f"[own branches] critic gradient elements outside the forced-branch bound: {outside} (what {flips['fake'] + flips['real'] + flips['hat']} "
          f"flipped critic units can touch: {d_budget}); generator worst rel-L2 {worst_g:.1e}")
    assert outside <= d_budget, (outside, d_budget)
    assert worst_g <= 2e-2, worst_g                        # BatchNorm's backward spreads one flipped unit over its whole channel



def test_celeba64_batch256_step_matches_oracle():
    """BASELINE.json configs[1], the headline: 64x64 (build-defined 64-arch), batch 256, sigma 5 -> 31 taps."""
    _full_batch_step_matches_oracle("celeba64", 256, 5.0)


def test_celeba128_batch128_step_matches_oracle():
    """BASELINE.json configs[3] (C4) at ITS batch: the verbatim 128-pixel stacks of demo_celeba.py:51-124, batch 128, sigma 5 ->
    31 taps, wgan.py:132-172,234-285.  At batch 128 the dispatcher picks other tiles, split-K plans, position-major orders and
    strip counts than at the batch 2-4 of the other celeba128 cases, and the 16-channel kernels' persistent strips run at their
    real occupancy: this is the one BASELINE configuration whose product dispatch had never met the oracle."""
    _full_batch_step_matches_oracle("celeba128", 128, 5.0)


def test_gradient_penalty_value_function():
    """The reference's module-level gradient_penalty(discriminator, reals, fakes) (wgan.py:234-246), value only, on the HIP
    kernels (critic forward / data gradient / blur^T / per-sample norms / the loss kernel's mean) against the oracle."""
    from blurred_gan_amd.wgan import gradient_penalty
    arch, B = "tiny", 5
    gan, st, reals, rng = _make(arch, B, 1.2)
    fakes = rng.uniform(-1, 1, size=reals.shape)
    alpha = rng.uniform(size=B)
    want = S.gradient_penalty(st, reals, fakes, alpha, want_grads=False)[0]
    got = float(gradient_penalty(gan.discriminator, torch.from_numpy(reals), torch.from_numpy(fakes), torch.from_numpy(alpha)))
    assert abs(got - want) < 1e-4 * max(1.0, abs(want)), (got, want)


@pytest.mark.parametrize("arch,B", [("tiny", 5), ("mnist", 6), ("celeba64", 4)])
def test_merged_penalty_filter_gradients_equal_the_separate_launches(arch, B):
    """engine.Net.gp_second_order_merged (the penalty's second-order filter gradients taken in the SAME launches as the merged
    critic pass's own, over all 3B rows) against the separate launches it replaced (WGAN(merge_gp_filter_gradients=False)), and
    against the two-pass critic (merge_critic_passes=False): the same critic gradients up to summation order."""
    ref = None
    for kw in (dict(), dict(merge_gp_filter_gradients=False), dict(merge_critic_passes=False)):
        gan, st, reals, rng = _make(arch, B, 1.0, seed=4, **kw)
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        gan.discriminator.optimizer.learning_rate = 0.0
        gan.generator.optimizer.learning_rate = 0.0
        gan.train_on_batch(reals.astype(np.float32), randomness=rnd)
        grads = product_grads(gan.discriminator)
        if ref is None:
            ref = grads
            continue
        for a, b in zip(grads, ref):
            assert rel_l2(a, b) < 2e-6 or not np.any(b), (kw, a.shape, rel_l2(a, b))


@pytest.mark.parametrize("arch,B", [("celeba64", 64), ("celeba128", 4)])
def test_fused_batchnorm_statistics_equal_the_separate_pass(arch, B):
    """BatchNorm batch statistics left by the producing conv's epilogue (engine.Net.fuse_bn_stats, opt-in since round 3:
    BGAN_FUSED_BN_STATS=1) against the separate statistics pass (the default): the same generator gradients and moving statistics up
    to summation order -- which the BatchNorm backward's cancellation amplifies to the level of the float32 oracle's own
    deviation from float64 (helpers.GRAD_L2["g"]): the bound on the gradients is that one, the statistics themselves agree to 1e-5."""
    ref = None
    for fused in (False, True):
        gan, st, reals, rng = _make(arch, B, 1.0, seed=6)
        gan.generator.net().fuse_bn_stats = fused
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        gan.discriminator.optimizer.learning_rate = 0.0
        gan.generator.optimizer.learning_rate = 0.0
        gan.train_on_batch(reals.astype(np.float32), randomness=rnd)
        grads = product_grads(gan.generator)
        moving = [v.detach().cpu().numpy().astype(np.float64) for l in gan.generator.layers
                  for k, v in getattr(l, "vars", {}).items() if k.startswith("moving_")]
        if ref is None:
            ref = (grads, moving)
            continue
        assert len(moving) == len(ref[1]) and len(moving) > 0
        for a, b in zip(moving, ref[1]):
            assert rel_l2(a, b) < 1e-5, (a.shape, rel_l2(a, b))
        for a, b in zip(grads, ref[0]):
            assert rel_l2(a, b) < 2e-3 and cosine(a, b) > 1 - 1e-5 or not np.any(b), (a.shape, rel_l2(a, b), cosine(a, b))


def test_vector_loss_quirk_switch():
    """Q1 on/off changes the critic gradient exactly by the documented factor on the W + GP part."""
    arch, B = "tiny", 4
    gan_q, st, reals, rng = _make(arch, B, 0.5, reproduce_vector_loss_quirk=True)
    gan_n, _, _, _ = _make(arch, B, 0.5, reproduce_vector_loss_quirk=False)
    rnd = S.draw_randomness(arch, B, rng, np.float64)
    for g in (gan_q, gan_n):
        g.discriminator.optimizer.learning_rate = 0.0
        g.generator.optimizer.learning_rate = 0.0
        g.hparams.e_drift = 0.0
        g.train_on_batch(reals.astype(np.float32), randomness=rnd)
    for a, b in zip(product_grads(gan_q.discriminator), product_grads(gan_n.discriminator)):
        np.testing.assert_allclose(a, B * b, rtol=2e-3, atol=1e-5 * max(1e-6, np.abs(a).max()))


def test_training_reduces_critic_loss_own_rng():
    """Plumbing run with the build's own RNG (no injected randomness): finite metrics, counters, sigma schedule."""
    import blurred_gan_amd as bg
    from blurred_gan_amd import models, callbacks
    bg.set_seed(123123)
    arch, B = "tiny", 8
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=1.0, global_batch_size=B, batch_size=B)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir="/tmp/bg_test_logs"))
    data = [torch.rand(B, 8, 8, 3) * 2 - 1 for _ in range(6)]
    hist = gan.fit(data, epochs=2, callbacks=[callbacks.BlurDecayController(total_n_training_examples=6 * B * 2, max_value=1.0)])
    assert int(gan.n_batches) == 12 and int(gan.n_img) == 12 * B
    assert all(np.isfinite(v) for v in hist[-1].values())
    exp = 1.0 * 0.96 ** (11 / (6 * B * 2 / 10))
    assert abs(float(gan.std) - exp) < 1e-5
    s = gan.generate_samples(training=False)
    assert tuple(s.shape) == (B, 8, 8, 3) and torch.isfinite(s).all() and s.abs().max() <= 1.0


@pytest.mark.parametrize("arch,B,std", [("tiny", 6, 1.0), ("mnist", 8, 0.05), ("celeba64", 8, 5.0)])
def test_loss_curves_track_the_oracle_over_25_steps(arch, B, std):
    """Loss-curve parity (BASELINE.json: 'loss curves within tolerance of the CPU reference'; the loop of wgan.py:86-114):
    25 consecutive FREE-RUNNING steps with injected randomness -- the product, the float64 oracle and the float32 oracle each
    carry their own weights, BN statistics and Adam slots forward; nothing is re-synchronised -- on the 8x8 test stack and on
    the real MNIST and 64x64 stacks.

    Criterion, per step t and metric k, e(t, k) = |metric - float64 oracle| / (0.1 + max_j |float64 oracle metric j at t|)
    -- the deviation against the SCALE of the step's metrics: a loss crossing zero while the scores are at 10 is not a 20 %
    error -- held to
        e_HIP(t, k) <= max(1e-2, 3 * max_{s <= t, j} e_float32-oracle(s, j)).
    The plain per-metric figure |d| / (|ref| + 0.1) of both is printed beside it.
    The plain 1 % bound alone is what the 8x8 stack is held to (and meets); at the real stacks it is not attainable by ANY
    float32 implementation of this loop: the trajectory is chaotic at batch 8 (losses swing between -120 and +190 at
    celeba64), and the ORACLE ITSELF run in float32 -- the reference's own precision, TF computes in float32 -- leaves its
    float64 run by 2.1 % (mnist) and 3.7 % (celeba64) within 25 steps (measured on the CPU, oracle against oracle).  So the
    test carries the float32 oracle along as the yardstick of the precision class and holds the HIP path to a small multiple
    of ITS deviation; the first steps, where conditioning is still good, are effectively held to the 1 % bound."""
    gan, st, reals, rng = _make(arch, B, std, seed=21)
    st32 = copy.deepcopy(st)
    for key in ("g", "d", "g_m", "g_v", "d_m", "d_v"):
        st32[key] = [{k: v.astype(np.float32) for k, v in p.items()} for p in st32[key]]
    hp = dict(S.DEFAULT_HP, global_batch_size=B)
    names = ("disc_loss", "gen_loss", "gp_term", "real_scores", "fake_scores")
    worst, worst32, env32, curve, within_1pct, raw, raw32 = 0.0, 0.0, 0.0, [], 0, 0.0, 0.0
    for it in range(25):
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        r = rng.uniform(-1, 1, size=reals.shape)
        st, met, _ = S.train_on_batch(st, r, rnd, hp)
        rnd32 = {k: (v.astype(np.float32) if isinstance(v, np.ndarray) else v) for k, v in rnd.items()}
        st32, met32, _ = S.train_on_batch(st32, r.astype(np.float32), rnd32, hp)
        got = dict(zip(gan.metrics_names, gan.train_on_batch(r.astype(np.float32), randomness=rnd)))
        scale = 0.1 + max(abs(met[k]) for k in names)
        e_hip = max(abs(got[k] - met[k]) for k in names) / scale
        e_32 = max(abs(met32[k] - met[k]) for k in names) / scale
        raw_hip = max(abs(got[k] - met[k]) / (abs(met[k]) + 0.1) for k in names)
        raw_32 = max(abs(met32[k] - met[k]) / (abs(met[k]) + 0.1) for k in names)
        raw, raw32 = max(raw, raw_hip), max(raw32, raw_32)
        env32 = max(env32, e_32)
        worst, worst32 = max(worst, e_hip), max(worst32, e_32)
        within_1pct += raw_hip < 1e-2
        curve.append((it, round(met["disc_loss"], 3), round(got["disc_loss"], 3), round(met["gen_loss"], 3), round(got["gen_loss"], 3),
                      f"{e_hip:.1e}", f"{e_32:.1e}"))
        assert e_hip <= max(1e-2, 3.0 * env32), (arch, it, e_hip, env32, curve)
    print(f"{arch} B={B}: (step, disc_loss oracle64 / HIP, gen_loss oracle64 / HIP, deviation HIP, deviation float32 oracle):", curve)
    print(f"{arch} B={B}: worst deviation over 25 free-running steps, against the step's metric scale: HIP {worst:.2e}, float32 oracle "
          f"{worst32:.2e}; per metric |d|/(|ref|+0.1): HIP {raw:.2e}, float32 oracle {raw32:.2e}; steps with HIP inside the plain 1 % "
          f"per-metric bound: {within_1pct}/25")
    assert int(gan.n_batches) == 25
    if arch == "tiny":
        assert raw < 1e-2


def test_resume_from_checkpoint_equals_uninterrupted_run(tmp_path):
    """N2 (demo_mnist.py:145-163,191): 3 steps, save, restore into a NEW model, 3 more steps == 6 uninterrupted steps, bit
    for bit -- weights, BN statistics, Adam slots, counters, sigma AND the positions of the build's own random streams
    (latents, alpha, dropout masks), with the step's own RNG (nothing injected)."""
    import blurred_gan_amd as bg
    from blurred_gan_amd import models, callbacks
    from blurred_gan_amd.checkpoint import CheckpointManager
    arch, B = "mnist", 8

    def make(seed):
        bg.set_seed(seed)
        gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
        hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=2.0, global_batch_size=B, batch_size=B)
        return bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir=str(tmp_path / "log")))
    g = torch.Generator().manual_seed(1)
    data = [torch.rand(B, 28, 28, 1, generator=g) * 2 - 1 for _ in range(6)]
    ctl = lambda: callbacks.BlurDecayController(total_n_training_examples=100, max_value=2.0)
    full = make(7)
    full.fit(data, epochs=1, callbacks=[ctl()])
    first = make(7)
    first.fit(data[:3], epochs=1, callbacks=[ctl()])
    mgr = CheckpointManager(first, str(tmp_path / "ckpt"))
    mgr.save()
    resumed = make(99)                                    # different initial weights and seed: everything must come from the file
    CheckpointManager(resumed, str(tmp_path / "ckpt")).restore(mgr.latest_checkpoint)
    assert int(resumed.n_batches) == 3 and int(resumed.n_img) == 3 * B
    resumed.fit(data[3:], epochs=1, callbacks=[ctl()])
    assert int(resumed.n_batches) == int(full.n_batches) == 6 and float(resumed.std) == float(full.std)
    for a, b in ((resumed.generator, full.generator), (resumed.discriminator, full.discriminator)):
        for name in ("theta", "state", "m", "v"):
            assert torch.equal(getattr(a.store, name), getattr(b.store, name)), name
        assert a.optimizer.iterations == b.optimizer.iterations == 6
