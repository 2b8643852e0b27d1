"""N3: the build's SWD implementation against outputs of the REFERENCE sliced_wasserstein.py stored in
tests/golden/swd_golden.npz (made by tests/golden/make_swd_golden.py), and the Frechet-distance formula against
closed forms."""
import os

import numpy as np
import pytest

from blurred_gan_amd import metrics, sliced_wasserstein as sw

G = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "swd_golden.npz"))


def test_pyramid_steps_match_reference():
    np.testing.assert_allclose(sw.pyr_up(G["small"]), G["small_up"], rtol=1e-5, atol=1e-6)
    np.testing.assert_allclose(sw.pyr_down(G["down_in"]), G["small_down"], rtol=1e-5, atol=1e-6)
    pyr = sw.generate_laplacian_pyramid(G["batch"], 2)
    np.testing.assert_allclose(pyr[0], G["pyr0"], rtol=1e-4, atol=2e-3)
    np.testing.assert_allclose(pyr[1], G["pyr1"], rtol=1e-5, atol=1e-3)
    rec = sw.reconstruct_laplacian_pyramid(pyr)
    np.testing.assert_allclose(rec, G["batch"], rtol=1e-4, atol=2e-3)       # the pyramid is invertible


def test_pyramid_does_not_mutate_float32_input():
    x0 = G["batch"].astype(np.float32)
    x = x0.copy()
    sw.generate_laplacian_pyramid(x, 2)
    np.testing.assert_array_equal(x, x0)                                     # the reference mutates float32 input


def test_descriptors_and_finalize_match_reference_with_same_seed():
    desc = sw.get_descriptors_for_minibatch(G["level"], 7, 5, np.random.RandomState(4321))
    np.testing.assert_array_equal(desc, G["desc"])
    np.testing.assert_allclose(sw.finalize_descriptors(desc), G["desc_final"], rtol=1e-5, atol=1e-6)


def test_sliced_wasserstein_matches_reference_with_same_seed():
    got = sw.sliced_wasserstein(G["A"], G["B"], 3, 16, np.random.RandomState(999))
    assert abs(got - float(G["swd"])) < 1e-6 * max(1.0, abs(float(G["swd"])))
    # properties: identical sets -> 0; symmetric
    assert sw.sliced_wasserstein(G["A"], G["A"], 2, 8, np.random.RandomState(1)) == 0.0
    ab = sw.sliced_wasserstein(G["A"], G["B"], 2, 8, np.random.RandomState(5))
    ba = sw.sliced_wasserstein(G["B"], G["A"], 2, 8, np.random.RandomState(5))
    assert abs(ab - ba) < 1e-7


def test_api_end_to_end_matches_reference():
    api = sw.API((4, 32, 32, 3), seed=2024)
    api.begin("reals"); api.feed("reals", G["api_reals"]); api.end("reals")
    api.begin("fakes"); api.feed("fakes", G["api_fakes"]); res = api.end("fakes")
    np.testing.assert_allclose(res, G["api_result"], rtol=1e-4)
    assert api.get_metric_names() == ["SWDx1e3_32", "SWDx1e3_16", "SWDx1e3_avg"]


def test_swd_metric_object_and_reference_bug_switch():
    reals = G["api_reals"]
    ramp = np.linspace(0, 255, 32)[None, None, None, :] + np.linspace(0, 255, 32)[None, None, :, None]
    fakes = np.broadcast_to(ramp / 2, reals.shape) + np.random.RandomState(3).normal(size=reals.shape)   # smooth images: a different distribution
    m = metrics.SWDMetric(seed=0)
    m.update_state(reals, fakes)
    r = m.results()
    assert set(r) == {"SWDx1e3_32", "SWDx1e3_16", "SWDx1e3_avg"} and r["SWDx1e3_avg"] > 0
    assert abs(m.result() - r["SWDx1e3_avg"]) < 0.2 * r["SWDx1e3_avg"] and m.name == "SWDx1e3_avg"   # fresh random directions per call
    b = metrics.SWDMetric(seed=0, reproduce_reference_bug=True)      # metrics.py:131: fakes built from the REAL minibatch
    b.update_state(reals, fakes)
    assert b.result() < 0.7 * m.result()                             # real-vs-real sampling noise only
    m.reset_states()
    assert all(len(l) == 0 for l in m.real_descriptors)


def test_frechet_distance_closed_forms():
    rng = np.random.RandomState(0)
    x = rng.normal(size=(4000, 6))
    assert abs(metrics.calculate_fid_safe(x, x)) < 1e-6
    shift = np.array([1.0, -2.0, 0.5, 0, 0, 3.0])
    assert abs(metrics.calculate_fid_safe(x, x + shift) - shift.dot(shift)) < 1e-6       # equal covariances: ||dmu||^2
    y = 2.0 * x                                                                           # S2 = 4 S1: Tr(S1 + 4S1 - 4S1) = Tr(S1)
    s1 = np.cov(x, rowvar=False)
    mu = x.mean(0)
    assert abs(metrics.calculate_fid_safe(x, y) - (mu.dot(mu) + np.trace(s1))) < 1e-5
    fid = metrics.FIDMetric(feature_extractor=lambda imgs: imgs.reshape(len(imgs), -1)[:, :6])
    fid.update_state(x.reshape(4000, 6, 1, 1), (x + shift).reshape(4000, 6, 1, 1))
    assert abs(fid.result() - shift.dot(shift)) < 1e-6
