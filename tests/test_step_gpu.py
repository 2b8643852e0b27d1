"""GPU parity of the whole hot path: BlurredWGANGP.train_on_batch (through the reference-shaped Python API and
the C ABI) vs the numpy oracle, on identical weights and injected randomness."""
import copy

import numpy as np
import pytest
import torch

from oracle import step as S
from helpers import load_oracle_weights, product_grads, oracle_grad_list, oracle_weight_list

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
    pg = product_grads(gan.discriminator)
    for a, b in zip(pg, oracle_grad_list(dg)):
        scale = max(np.abs(b).max(), 1e-6)
        np.testing.assert_allclose(a, b.reshape(a.shape), rtol=2e-3, atol=2e-4 * scale)
    gg, upd, gm = S.generator_grads(st, rnd, hp, B)
    pgg = product_grads(gan.generator)
    # Generator gradients pass through five BatchNorm backwards (dz - mean(dz) - xhat*mean(dz*xhat)): at batch 64 the
    # cancellation leaves gradients of 1e-5 whose fp32 evaluation is itself only good to ~1e-2 of their maximum -- the ORACLE
    # run in float32 deviates from its float64 run by 5e-4...1.4e-2 on this case (the HIP path: 1e-4...3e-3).
    g_atol = 1e-2 if B >= 64 else 2e-4
    for a, b in zip(pgg, oracle_grad_list(gg)):
        scale = max(np.abs(b).max(), 1e-6)
        np.testing.assert_allclose(a, b.reshape(a.shape), rtol=2e-3, atol=g_atol * scale)
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


def test_loss_curves_track_the_oracle_over_25_steps():
    """Loss-curve parity (BASELINE.json: 'loss curves within tolerance of the CPU reference'): 25 consecutive steps with
    injected randomness; every per-step metric of the HIP path stays within 1 % (+1e-3) of the float64 oracle's."""
    arch, B = "tiny", 6
    gan, st, reals, rng = _make(arch, B, 1.0, seed=21)
    hp = dict(S.DEFAULT_HP, global_batch_size=B)
    worst = 0.0
    for it in range(25):
        rnd = S.draw_randomness(arch, B, rng, np.float64)
        r = rng.uniform(-1, 1, size=reals.shape)
        st, met, _ = S.train_on_batch(st, r, rnd, hp)
        got = dict(zip(gan.metrics_names, gan.train_on_batch(r.astype(np.float32), randomness=rnd)))
        for k in ("disc_loss", "gen_loss", "gp_term", "real_scores", "fake_scores"):
            err = abs(got[k] - met[k]) / (abs(met[k]) + 0.1)
            worst = max(worst, err)
            assert err < 1e-2, (it, k, got[k], met[k])
    assert int(gan.n_batches) == 25
    print("worst relative metric deviation over 25 steps:", worst)
