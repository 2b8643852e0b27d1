"""Evaluation metrics, host side: the Frechet distance of reference metrics.py:42-75 and the
SWD / FID metric objects its callbacks feed (metrics.py:93-184).  The reference's FID feature extractor is an
Inception-v3 fetched from tfhub.dev (metrics.py:169-170) -- unavailable offline -- so ``FIDMetric`` takes the feature
extractor as an argument; only the distance formula is pinned here."""
from __future__ import annotations

from typing import Dict, List

import numpy as np
from scipy.linalg import sqrtm

from . import sliced_wasserstein as sw


def calculate_fid_safe(act1: np.ndarray, act2: np.ndarray, epsilon=1e-6) -> float:
    """||mu1 - mu2||^2 + Tr(S1 + S2 - 2 sqrt(S1 S2)) with the singular-product fallback of metrics.py:42-75."""
    mu1, mu2 = np.atleast_1d(act1.mean(axis=0)), np.atleast_1d(act2.mean(axis=0))
    s1, s2 = np.atleast_2d(np.cov(act1, rowvar=False)), np.atleast_2d(np.cov(act2, rowvar=False))
    assert mu1.shape == mu2.shape and s1.shape == s2.shape
    covmean, _ = sqrtm(s1.dot(s2), disp=False)
    if not np.isfinite(covmean).all():
        off = np.eye(s1.shape[0]) * epsilon
        covmean = sqrtm((s1 + off).dot(s2 + off))
    if np.iscomplexobj(covmean):
        if not np.allclose(np.diagonal(covmean).imag, 0, atol=1e-3):
            raise ValueError("Imaginary component {}".format(np.max(np.abs(covmean.imag))))
        covmean = covmean.real
    d = mu1 - mu2
    return float(d.dot(d) + np.trace(s1) + np.trace(s2) - 2 * np.trace(covmean))


def _nchw_uint_like(x, on_device=False):
    if hasattr(x, "detach"):
        if on_device and x.is_cuda:
            return x.detach().to(sw.torch.float32)
        return x.detach().cpu().numpy()
    return np.asarray(x)


class SWDMetric:
    """metrics.py:93-157.  The reference builds the *fake* descriptors from the real minibatch (metrics.py:131) and never
    sets ``name`` (metrics.py:98); both are fixed here, ``reproduce_reference_bug=True`` restores the former."""

    def __init__(self, name="SWDx1e3_avg", dtype=None, seed=None, reproduce_reference_bug=False, on_device=False):
        """``on_device``: minibatches that arrive as tensors on the GPU stay there (the device path of sliced_wasserstein.py:
        same draws, float32 rounding apart); the default copies them to the host as the reference's callbacks do."""
        self.name = name
        self.on_device = on_device
        self.nhood_size, self.nhoods_per_image, self.dir_repeats, self.dirs_per_repeat = 7, 128, 4, 128
        self.resolutions: List[int] = []
        self.rng = np.random.RandomState(seed)
        self.reproduce_reference_bug = reproduce_reference_bug

    def get_metric_names(self):
        return ["SWDx1e3_%d" % r for r in self.resolutions] + ["SWDx1e3_avg"]

    def reset_states(self):
        for lst in getattr(self, "real_descriptors", []) + getattr(self, "fake_descriptors", []):
            lst.clear()

    def update_state(self, real_minibatch, fake_minibatch, *args, **kwargs):
        """Minibatches are NCHW with 3 channels (the callbacks' preprocessing converts, demo_mnist.py:180-184)."""
        real, fake = _nchw_uint_like(real_minibatch, self.on_device), _nchw_uint_like(fake_minibatch, self.on_device)
        if not self.resolutions:
            res = real.shape[2]
            while res >= 16:
                self.resolutions.append(res)
                res //= 2
            self.real_descriptors = [[] for _ in self.resolutions]
            self.fake_descriptors = [[] for _ in self.resolutions]
        n = len(self.resolutions)
        for lod, level in enumerate(sw.generate_laplacian_pyramid(real, n)):
            self.real_descriptors[lod].append(sw.get_descriptors_for_minibatch(level, self.nhood_size, self.nhoods_per_image, self.rng))
        src = real if self.reproduce_reference_bug else fake
        for lod, level in enumerate(sw.generate_laplacian_pyramid(src, n)):
            self.fake_descriptors[lod].append(sw.get_descriptors_for_minibatch(level, self.nhood_size, self.nhoods_per_image, self.rng))

    def results(self) -> Dict[str, float]:
        dr = [sw.finalize_descriptors(d) for d in self.real_descriptors]
        df = [sw.finalize_descriptors(d) for d in self.fake_descriptors]
        dist = [sw.sliced_wasserstein(a, b, self.dir_repeats, self.dirs_per_repeat, self.rng) * 1e3 for a, b in zip(dr, df)]
        dist.append(float(np.mean(dist)))
        return dict(zip(self.get_metric_names(), dist))

    def result(self):
        return self.results()[self.get_metric_names()[-1]]


class FIDMetric:
    """metrics.py:160-184 with the feature extractor injected (any callable images -> [N, D] features)."""

    def __init__(self, feature_extractor, name="FID"):
        self.name = name
        self.feature_extractor = feature_extractor
        self.reals: List[np.ndarray] = []
        self.fakes: List[np.ndarray] = []

    def update_state(self, real_minibatch, fake_minibatch, *args, **kwargs):
        self.reals.append(_nchw_uint_like(real_minibatch))
        self.fakes.append(_nchw_uint_like(fake_minibatch))

    def reset_states(self):
        self.reals.clear()
        self.fakes.clear()

    def result(self):
        fr = np.asarray(self.feature_extractor(np.concatenate(self.reals, 0)))
        ff = np.asarray(self.feature_extractor(np.concatenate(self.fakes, 0)))
        return calculate_fid_safe(fr, ff)
