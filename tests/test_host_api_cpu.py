"""CPU tests of the reference-shaped host API (no device compute): layer shape inference with the reference's
own asserts, parameter counts, hyper-parameter dataclasses / CLI / JSON, callbacks' sigma schedule, fit protocol."""
import argparse
import json
import os

import numpy as np
import pytest

import blurred_gan_amd as bg
from blurred_gan_amd import callbacks, layers, models


def test_generator_shapes_follow_reference_asserts():
    g = models.DCGANGenerator(arch="celeba128")            # constructor runs demo_celeba.py:60-93's asserts
    assert g.output_shape == (None, 128, 128, 3) and g.input_shape == (None, 100)
    assert models.DCGANGenerator(arch="mnist").output_shape == (None, 28, 28, 1)
    assert models.DCGANGenerator(arch="celeba64").output_shape == (None, 64, 64, 3)
    d = models.DCGANDiscriminator(arch="celeba128")
    assert d.input_shape == (None, 128, 128, 3) and d.output_shape == (None, 1)


@pytest.mark.parametrize("arch,g_n,d_n", [("celeba128", 11738800, 4368048), ("celeba64", 11727200, 4356448), ("mnist", 2280000, 212672)])
def test_param_counts(arch, g_n, d_n):
    g, d = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    kern = lambda m: sum(int(l.vars["kernel"].numel()) for l in m.build()._own_layers() if "kernel" in l.vars)
    assert kern(g) == g_n and kern(d) == d_n
    assert g.count_params() > g_n and len(d.trainable_variables) == 2 * (len(models._D[arch]) + 1)


def test_blurred_variant_wraps_critic_and_shares_variables():
    g, d = models.DCGANGenerator(arch="tiny"), models.DCGANDiscriminator(arch="tiny")
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=1.5)
    gan = bg.BlurredWGANGP(g, d, hp, bg.TrainingConfig())
    assert gan.discriminator is not d and gan.discriminator.layers[0] is gan.blur and gan.discriminator.layers[1] is d
    assert float(gan.std) == 1.5
    gan.std.assign(0.25)
    assert float(gan.blur.std) == 0.25
    # inner model's variables alias the wrapper's store
    d.trainable_variables[0].fill_(3.0)
    assert float(gan.discriminator.trainable_variables[0].flatten()[0]) == 3.0
    assert gan.metrics_names == ["loss", "real_scores", "fake_scores", "gen_loss", "disc_loss", "gp_term", "norm_term", "std"]
    assert bg.BlurredWGAN.__name__ == "BlurredWGAN" and issubclass(bg.BlurredWGANGP, bg.WGANGP)
    net = gan.discriminator.net()
    assert net.blur is gan.blur and [s.kind for s in net.stages] == ["conv", "conv", "dense"]
    assert all(s.act == "lrelu" and s.drop == 0.3 for s in net.stages[:2])


def test_hyperparameter_defaults_cli_and_json(tmp_path):
    hp = bg.BlurredWGANGP.HyperParameters()
    assert (hp.learning_rate, hp.d_steps_per_g_step, hp.batch_size, hp.global_batch_size, hp.optimizer) == (0.001, 1, 32, 32, "adam")
    assert (hp.e_drift, hp.gp_coefficient, hp.initial_blur_std) == (1e-4, 10.0, 0.05)
    cfg = bg.TrainingConfig()
    assert (cfg.log_dir, cfg.checkpoint_dir, cfg.save_image_summaries_interval) == ("results/log", "results/log/checkpoints", 50)
    p = argparse.ArgumentParser()
    bg.BlurredWGANGP.HyperParameters.add_arguments(p)
    bg.TrainingConfig.add_arguments(p)
    a = p.parse_args(["--learning_rate", "0.01", "--global_batch_size", "256", "--log_dir", "x/y"])
    hp2, cfg2 = bg.BlurredWGANGP.HyperParameters.from_args(a), bg.TrainingConfig.from_args(a)
    assert hp2.learning_rate == 0.01 and hp2.global_batch_size == 256 and cfg2.log_dir == "x/y"
    f = tmp_path / "hp.json"
    hp2.save_json(str(f))
    assert json.load(open(f))["gp_coefficient"] == 10.0
    assert bg.BlurredWGANGP.HyperParameters.from_json(str(f)) == hp2


def test_blur_decay_controller_schedule_q5():
    class M:
        pass
    m = M()
    m.n_batches, m.std = 0, bg.gaussian_blur.Variable(0.0)
    ctl = callbacks.BlurDecayController(total_n_training_examples=600000, max_value=5, min_value=0.01)
    ctl.set_model(m)
    for nb, exp in [(0, 5.0), (60000, 5.0 * 0.96), (18750, 5.0 * 0.96 ** 0.3125)]:
        m.n_batches = nb
        ctl.on_batch_begin(0, {})
        assert abs(float(m.std) - exp) < 1e-5


def test_execute_every_n_examples_counts():
    calls = []

    class C(callbacks.ExecuteEveryNExamplesCallback):
        def function(self, batch, logs):
            calls.append(self.samples_seen)
    c = C(n=100)
    for b in range(10):
        c.on_batch_end(b, {"size": 32})
    assert calls == [32, 128, 224, 320]


def test_execute_every_n_examples_start_offset_and_catch_up():
    """Nothing before ``starting_from``; a negative offset makes the first batch due; one call per batch while catching up."""
    def run(n, start, sizes):
        calls = []

        class C(callbacks.ExecuteEveryNExamplesCallback):
            def function(self, batch, logs):
                calls.append((batch, self.samples_seen))
        c = C(n=n, starting_from=start)
        for b, sz in enumerate(sizes):
            c.on_batch_end(b, {"size": sz})
        return calls
    assert run(100, 250, [64] * 8) == [(3, 256), (5, 384), (7, 512)]
    assert run(16, -16, [8, 8, 8, 8]) == [(0, 8), (1, 16), (2, 24), (3, 32)]   # a head start of one period: a call every batch
    assert run(10, 0, [35, 1, 1, 1, 1]) == [(0, 35), (1, 36), (2, 37), (3, 38)]
    with pytest.raises(NotImplementedError):
        callbacks.ExecuteEveryNExamplesCallback(5).on_batch_end(0, {"size": 8})


def test_adaptive_blur_controller_logs_and_stops():
    """callbacks.py:65-135: EMA of the fake share of the scores, silent during warm-up, one multiplicative step per 100 batches
    while balanced (logged as would_modify, never assigned to the model), training stops under min_value."""
    logged = []

    class W:
        def as_default(self):
            import contextlib
            return contextlib.nullcontext(self)

        def scalar(self, name, value, step=None):
            logged.append((name, value))

    class M:
        summary_writer, stop_training = W(), False
    m = M()
    m.std = bg.gaussian_blur.Variable(0.0)
    ctl = callbacks.AdaptiveBlurController(smoothing=0.5, warmup_n_batches=3, threshold=0.05, min_value=1.0, max_value=4.0)
    ctl.set_model(m)
    ctl.on_train_begin()
    assert float(m.std) == 4.0
    for b in range(3):                                     # warm-up: the average moves, nothing is logged
        ctl.on_batch_end(b, {"fake_scores": 1.0, "real_scores": 3.0})
    assert not logged and abs(ctl.score_ratio - (0.25 + 0.25 * 0.5 ** 3)) < 1e-12 and not ctl.gan_problem_is_stable()
    ctl.on_batch_end(3, {"fake_scores": 1.0, "real_scores": 3.0})
    assert [n for n, _ in logged] == ["blur_controller/ratio", "blur_controller/smoothed_ratio", "blur_controller/stable"]
    assert logged[-1][1] == 0 and ctl.std == 4.0
    logged.clear()
    ctl.score_ratio = 0.5
    for b in (50, 120, 180, 230, 400):                     # balanced: steps at 120 (>= 100 after 0), 230, 400 -- not at 50 / 180
        ctl.on_batch_end(b, {"fake_scores": 2.0, "real_scores": 2.0})
    mods = [v for n, v in logged if n.endswith("would_modify")]
    assert mods == [0, 1, 0, 1, 1] and ctl.std == 4.0 * 0.5 ** 3 and float(m.std) == 4.0
    assert m.stop_training                                 # 0.5 < min_value


def test_unsupported_layer_patterns_raise():
    s = layers.Sequential([layers.Dense(4, input_shape=(3,)), layers.LeakyReLU()])
    with pytest.raises(NotImplementedError):
        s.net()
    with pytest.raises(NotImplementedError):
        layers.Conv2D(4, 3, padding="valid")
    with pytest.raises(NotImplementedError):
        bg.WGAN(models.DCGANGenerator(arch="tiny"), models.DCGANDiscriminator(arch="tiny"),
                bg.WGAN.HyperParameters(optimizer="sgd"), bg.TrainingConfig())


def test_gaussian_blur_module_helpers():
    gb = bg.gaussian_blur
    assert gb.appropriate_kernel_size(5.0) == 31 and gb.appropriate_std(31) == 5.0
    assert abs(gb.maximum_reasonable_std(256) - 254 / 6) < 1e-9
    import torch
    assert gb.get_data_format(torch.zeros(1, 8, 8, 3)) == "NHWC" and gb.get_data_format(torch.zeros(1, 16, 8, 8)) == "NCHW"
    assert gb.get_image_dims(torch.zeros(2, 5, 7, 3)) == (5, 7, 3)


def test_checkpoint_manager_round_trip_and_rotation(tmp_path):
    from blurred_gan_amd.checkpoint import CheckpointManager
    import torch

    def make():
        bg.set_seed(5)
        g, d = models.DCGANGenerator(arch="tiny"), models.DCGANDiscriminator(arch="tiny")
        return bg.BlurredWGANGP(g, d, bg.BlurredWGANGP.HyperParameters(initial_blur_std=1.5), bg.TrainingConfig())
    gan = make()
    mgr = CheckpointManager(gan, str(tmp_path / "ckpt"), max_to_keep=2)
    assert mgr.latest_checkpoint is None
    gan.n_img.assign(640); gan.n_batches.assign(20); gan.std.assign(0.75)
    gan.generator.store.ensure_opt_state()
    gan.generator.store.m.fill_(0.25)
    gan.generator.optimizer.iterations = 7
    gan.generator.trainable_variables[0].fill_(1.5)
    for n in (100, 200, 300):
        mgr.save(n)
    assert [p.split("-")[-1] for p in mgr.checkpoints] == ["200.npz", "300.npz"]          # max_to_keep
    gan._rng_off = 1234
    gan.discriminator.net().rng_offset = 77
    mgr.save(300)
    other = make()
    CheckpointManager(other, str(tmp_path / "ckpt")).restore(mgr.latest_checkpoint)
    assert int(other.n_img) == 640 and int(other.n_batches) == 20 and abs(float(other.std) - 0.75) < 1e-7
    assert other.generator.optimizer.iterations == 7
    assert torch.equal(other.generator.store.theta, gan.generator.store.theta)
    assert torch.equal(other.generator.store.m, gan.generator.store.m)
    assert torch.equal(other.discriminator.store.state, gan.discriminator.store.state)
    assert other._rng_off == 1234 and other.discriminator.net().rng_offset == 77      # random streams continue, not replay


def test_checkpoint_manager_orders_by_save_time_not_by_file_number(tmp_path):
    """tf.train.CheckpointManager semantics: SaveModelCallback numbers files with a counter that restarts at every fit()
    (callbacks.py:245), so after a resume the NEW files carry SMALLER numbers; they must still be the latest and the
    ones kept (save -> restore -> fit -> save)."""
    from blurred_gan_amd.checkpoint import CheckpointManager
    bg.set_seed(5)
    g, d = models.DCGANGenerator(arch="tiny"), models.DCGANDiscriminator(arch="tiny")
    gan = bg.BlurredWGANGP(g, d, bg.BlurredWGANGP.HyperParameters(), bg.TrainingConfig())
    mgr = CheckpointManager(gan, str(tmp_path / "c"), max_to_keep=2)
    gan.n_img.assign(100000)
    mgr.save(100000)
    mgr2 = CheckpointManager(gan, str(tmp_path / "c"), max_to_keep=2)      # the resumed process
    mgr2.restore(mgr2.latest_checkpoint)
    gan.n_img.assign(100032)
    mgr2.save(32)
    assert mgr2.latest_checkpoint.endswith("ckpt-32.npz")
    gan.n_img.assign(110016)
    mgr2.save(10016)
    assert [os.path.basename(p) for p in mgr2.checkpoints] == ["ckpt-32.npz", "ckpt-10016.npz"]
    assert not os.path.exists(tmp_path / "c" / "ckpt-100000.npz")
    assert not [n for n in os.listdir(tmp_path / "c") if "tmp" in n]       # atomic writes leave no partial files
    keep_all = CheckpointManager(gan, str(tmp_path / "k"), max_to_keep=None)
    for n in range(7):
        keep_all.save(n)
    assert len(keep_all.checkpoints) == 7


def test_checkpoint_manager_ignores_and_removes_temp_files_of_a_crashed_save(tmp_path):
    """A save that died mid-write leaves ckpt-N.npz.tmp.npz behind; in a directory without a readable index it must not
    become latest_checkpoint (it is the newest file there), and the next save() removes it."""
    from blurred_gan_amd.checkpoint import CheckpointManager
    bg.set_seed(5)
    g, d = models.DCGANGenerator(arch="tiny"), models.DCGANDiscriminator(arch="tiny")
    gan = bg.BlurredWGANGP(g, d, bg.BlurredWGANGP.HyperParameters(), bg.TrainingConfig())
    mgr = CheckpointManager(gan, str(tmp_path / "c"))
    mgr.save(10)
    os.remove(mgr._index_path())                                       # index lost / written by hand
    with open(tmp_path / "c" / "ckpt-20.npz.tmp.npz", "wb") as f:      # truncated temp file, newest in the directory
        f.write(b"PK\x03\x04 truncated")
    fresh = CheckpointManager(gan, str(tmp_path / "c"))
    assert fresh.latest_checkpoint.endswith("ckpt-10.npz")
    fresh.restore(fresh.latest_checkpoint)
    fresh.save(30)
    assert sorted(os.listdir(tmp_path / "c")) == ["checkpoint.json", "ckpt-10.npz", "ckpt-30.npz"] or \
        not [n for n in os.listdir(tmp_path / "c") if n.endswith(".tmp.npz")]


def test_feed_images_to_metric_callback_counts():
    class M:
        name = "m"
        def __init__(self): self.n = 0
        def update_state(self, r, f): self.n += len(r)
        def result(self): return float(self.n)
        def reset_states(self): self.n = 0
    import torch
    from blurred_gan_amd.wgan import _SummaryWriter
    class Model:
        pass
    model = Model()
    model.images = (torch.zeros(32, 2), torch.zeros(32, 2))
    model.n_img = 0
    model.summary_writer = _SummaryWriter("/tmp/bg_cb_logs")
    cb = callbacks.FeedImagesToMetricCallback(M(), lambda x: x, num_samples=50, every_n_examples=100)
    cb.set_model(model)
    for b in range(8):
        cb.on_batch_end(b, {"size": 32})
    assert cb.results == [50.0, 50.0]        # recorded exactly num_samples each time (32 + 18), reference callbacks.py:156-173


def test_utils_file_and_image_helpers(tmp_path):
    """reference utils.py:27-103: run/epoch parsing, newest model file of the latest run, layout helpers, sample grid."""
    import dataclasses
    import numpy as np
    import torch
    from blurred_gan_amd import utils
    assert utils.run_id("results/07-mnist/model_12.hdf5") == 7
    assert utils.epoch("results/07-mnist/model_12.hdf5") == 12
    for run, eps in (("01-mnist", (1, 30)), ("03-mnist", (2, 11, 9)), ("02-celeba", (50,))):
        os.makedirs(tmp_path / run)
        for e in eps:
            (tmp_path / run / f"model_{e}.hdf5").write_text("")
    assert utils.locate_model_file(str(tmp_path), "mnist").endswith("03-mnist/model_11.hdf5")
    assert utils.locate_model_file(str(tmp_path), "celeba").endswith("02-celeba/model_50.hdf5")
    with pytest.raises(FileNotFoundError):
        utils.locate_model_file(str(tmp_path), "lsun")
    x = torch.arange(2 * 3 * 4 * 5).reshape(2, 3, 4, 5)
    assert utils.NHWC_to_NCHW(x).shape == (2, 5, 3, 4)
    assert torch.equal(utils.NCHW_to_NHWC(utils.NHWC_to_NCHW(x)), x)
    assert np.array_equal(utils.NCHW_to_NHWC(utils.NHWC_to_NCHW(x.numpy())), x.numpy())
    s = np.random.default_rng(0).uniform(size=(70, 4, 6, 3)).astype(np.float32)
    g = utils.samples_grid(s)
    assert g.shape == (32, 48, 3)
    assert np.array_equal(g[4:8, 12:18], s[1 * 8 + 2])            # row 1, column 2 of the grid is sample 10
    assert utils.samples_grid(s[..., :1]).shape == (32, 48)
    img = utils.plot_to_image(g)
    assert img.shape == (1, 32, 48, 4) and img.dtype == np.uint8 and (img[..., 3] == 255).all()
    ds = utils.to_dataset(s)
    assert len(ds) == 70 and np.array_equal(next(iter(ds)), s[0])
    it = iter([1, 2])
    assert utils.to_dataset(it) is it

    @dataclasses.dataclass
    class HP(utils.HyperParams):
        lr: float = 1e-3
        n: int = 5
    hp = HP(n=7)
    assert str(hp) == str({"lr": 1e-3, "n": 7})
    hp.save_json(str(tmp_path / "hp.json"))
    assert HP.from_json(str(tmp_path / "hp.json")) == hp
