"""N3 on the GPU: the metric feeders of reference callbacks.py:138-206 driven by a real `fit()` of the HIP path --
`model.images` (the fakes and reals of the D-step, wgan.py:103) go through the demo's preprocessing (normalise to [0, 1],
grayscale -> RGB, NHWC -> NCHW, demo_mnist.py:180-184) into SWDMetric / FIDMetric, and the recorded results equal the metrics
recomputed from those very images with the same seeds.  (The Frechet distance's feature extractor is injected: the reference's
Inception-v3 is a tfhub download.)"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _preprocess(images):
    from blurred_gan_amd import utils
    x = utils.normalize_images(images)                      # [-1, 1] -> [0, 1]  (utils.py:50-52)
    if x.shape[-1] == 1:
        x = x.repeat(1, 1, 1, 3)                            # tf.image.grayscale_to_rgb
    return utils.NHWC_to_NCHW(x) * 255.0                    # the SWD code works on 0..255 NCHW minibatches


def test_swd_and_fid_feeders_on_a_real_fit(tmp_path):
    import blurred_gan_amd as bg
    from blurred_gan_amd import models, callbacks, metrics, sliced_wasserstein as sw
    bg.set_seed(11)
    arch, B, nb = "mnist", 8, 4
    gen, disc = models.DCGANGenerator(arch=arch), models.DCGANDiscriminator(arch=arch)
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=1.0, global_batch_size=B, batch_size=B)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir=str(tmp_path / "log")))
    g = torch.Generator().manual_seed(5)
    data = [torch.rand(B, 28, 28, 1, generator=g) * 2 - 1 for _ in range(nb)]

    seen = []                                               # what the feeders saw, captured by a third callback

    class Tap(callbacks.Callback):
        def on_batch_end(self, batch, logs):
            fakes, reals = self.model.images
            seen.append((_preprocess(reals).cpu().numpy().copy(), _preprocess(fakes).cpu().numpy().copy()))

    proj = np.random.RandomState(0).normal(size=(3 * 28 * 28, 12))
    feats = lambda imgs: np.asarray(imgs, np.float64).reshape(len(imgs), -1) @ proj / 255.0
    swd_cb = callbacks.SWDMetricCallback(_preprocess, num_samples=2 * B, every_n_examples=2 * B, seed=3)
    fid_cb = callbacks.FIDMetricCallback(_preprocess, feats, num_samples=2 * B, every_n_examples=2 * B)
    gan.fit(data, epochs=1, callbacks=[Tap(), swd_cb, fid_cb])
    assert int(gan.n_batches) == nb and len(seen) == nb
    assert len(swd_cb.results) >= 1 and len(fid_cb.results) >= 1
    first = swd_cb.results[0]
    assert set(first) == {"SWDx1e3_28", "SWDx1e3_avg"} and all(np.isfinite(v) and v >= 0 for v in first.values())
    assert np.isfinite(fid_cb.results[0]) and fid_cb.results[0] >= 0
    # recompute from the captured images: the feeders start recording at the first batch (starting_from = -num_samples)
    m = metrics.SWDMetric(seed=3)
    f = metrics.FIDMetric(feats)
    for reals, fakes in seen[:2]:
        m.update_state(reals, fakes)
        f.update_state(reals, fakes)
    want = m.results()
    for k in first:
        assert abs(first[k] - want[k]) <= 1e-9 * max(1.0, abs(want[k])), (k, first[k], want[k])
    assert abs(fid_cb.results[0] - f.result()) <= 1e-9 * max(1.0, abs(f.result()))
    # real against real is (statistically) closer than real against fake for an untrained generator
    same = metrics.SWDMetric(seed=3)
    for reals, _ in seen[:2]:
        same.update_state(reals, reals)
    assert same.results()["SWDx1e3_avg"] < want["SWDx1e3_avg"]


def test_reference_golden_checks_on_the_gpu_box():
    """The one upstream-pinned number of the repository (tests/golden/swd_golden.npz: outputs of the REFERENCE
    sliced_wasserstein.py) re-checked where the product runs; tests/test_metrics_cpu.py is deselected by -m gpu."""
    import test_metrics_cpu as cpu
    cpu.test_pyramid_steps_match_reference()
    cpu.test_descriptors_and_finalize_match_reference_with_same_seed()
    cpu.test_sliced_wasserstein_matches_reference_with_same_seed()
    cpu.test_api_end_to_end_matches_reference()


def test_device_path_matches_the_reference_goldens():
    """N3 on the device: pyramid, descriptor gather, standardisation, projections and sort run on the card (torch tensors in,
    sliced_wasserstein.py's device branches), draws made on the host from the same RandomState -- held to the outputs of the
    REFERENCE module (tests/golden/swd_golden.npz), float32 rounding apart."""
    import test_metrics_cpu as cpu
    from blurred_gan_amd import sliced_wasserstein as sw
    G = cpu.G
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    np.testing.assert_allclose(sw.pyr_up(dev(G["small"])).cpu().numpy(), G["small_up"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sw.pyr_down(dev(G["down_in"])).cpu().numpy(), G["small_down"], rtol=1e-5, atol=1e-6)
    pyr = sw.generate_laplacian_pyramid(dev(G["batch"]), 2)
    assert all(p.is_cuda for p in pyr)
    np.testing.assert_allclose(pyr[0].cpu().numpy(), G["pyr0"], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(pyr[1].cpu().numpy(), G["pyr1"], rtol=1e-5, atol=1e-3)
    np.testing.assert_allclose(sw.reconstruct_laplacian_pyramid(pyr).cpu().numpy(), G["batch"], rtol=1e-4, atol=2e-3)
    desc = sw.get_descriptors_for_minibatch(dev(G["level"]), 7, 5, np.random.RandomState(4321))
    assert desc.is_cuda
    np.testing.assert_array_equal(desc.cpu().numpy(), G["desc"].astype(np.float32))        # a gather: exact
    np.testing.assert_allclose(sw.finalize_descriptors(desc).cpu().numpy(), G["desc_final"], rtol=1e-5, atol=2e-6)
    got = sw.sliced_wasserstein(dev(G["A"]), dev(G["B"]), 3, 16, np.random.RandomState(999))
    assert abs(got - float(G["swd"])) < 1e-5 * max(1.0, abs(float(G["swd"])))
    assert sw.sliced_wasserstein(dev(G["A"]), dev(G["A"]), 2, 8, np.random.RandomState(1)) == 0.0
    api = sw.API((4, 32, 32, 3), seed=2024)
    api.begin("reals"); api.feed("reals", dev(G["api_reals"])); api.end("reals")
    api.begin("fakes"); api.feed("fakes", dev(G["api_fakes"])); res = api.end("fakes")
    np.testing.assert_allclose(res, G["api_result"], rtol=1e-4)


def test_swd_metric_on_device_equals_host_metric():
    """SWDMetric(on_device=True) fed with GPU tensors against the host metric fed with the same images: same draws, same result to
    float32 rounding."""
    import test_metrics_cpu as cpu
    from blurred_gan_amd import metrics
    G = cpu.G
    reals = torch.from_numpy(np.ascontiguousarray(G["api_reals"], dtype=np.float32)).cuda()
    fakes = torch.from_numpy(np.ascontiguousarray(G["api_fakes"], dtype=np.float32)).cuda()
    md, mh = metrics.SWDMetric(seed=7, on_device=True), metrics.SWDMetric(seed=7)
    for m, r, f in ((md, reals, fakes), (mh, reals.cpu().numpy(), fakes.cpu().numpy())):
        m.update_state(r, f)
        m.update_state(r.flip(0) if hasattr(r, "flip") else r[::-1].copy(), f)
    assert all(d.is_cuda for lst in md.real_descriptors for d in lst)
    rd, rh = md.results(), mh.results()
    assert rd.keys() == rh.keys()
    for k in rd:
        assert abs(rd[k] - rh[k]) <= 1e-4 * max(1.0, abs(rh[k])), (k, rd[k], rh[k])


def test_swd_callback_on_device_in_a_real_fit(tmp_path):
    """SWDMetricCallback(on_device=True) beside the host one in the same fit(): same images, same seeds, same numbers to rounding."""
    import blurred_gan_amd as bg
    from blurred_gan_amd import models, callbacks
    bg.set_seed(3)
    B, nb = 8, 4
    gen, disc = models.DCGANGenerator(arch="mnist"), models.DCGANDiscriminator(arch="mnist")
    hp = bg.BlurredWGANGP.HyperParameters(initial_blur_std=1.0, global_batch_size=B, batch_size=B)
    gan = bg.BlurredWGANGP(gen, disc, hp, bg.TrainingConfig(log_dir=str(tmp_path / "log")))
    g = torch.Generator().manual_seed(2)
    data = [torch.rand(B, 28, 28, 1, generator=g) * 2 - 1 for _ in range(nb)]
    host = callbacks.SWDMetricCallback(_preprocess, num_samples=2 * B, every_n_examples=2 * B, seed=5)
    dev = callbacks.SWDMetricCallback(_preprocess, num_samples=2 * B, every_n_examples=2 * B, seed=5, on_device=True)
    gan.fit(data, epochs=1, callbacks=[host, dev])
    assert len(host.results) == len(dev.results) >= 1
    for rh, rd in zip(host.results, dev.results):
        for k in rh:
            assert abs(rh[k] - rd[k]) <= 1e-4 * max(1.0, abs(rh[k])), (k, rh[k], rd[k])
