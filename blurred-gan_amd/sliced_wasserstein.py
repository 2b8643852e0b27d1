"""Sliced Wasserstein distance between image sets: the evaluation metric the reference feeds from
``SWDMetricCallback`` (callbacks.py:186-198) -- Laplacian pyramid, 7x7x3 neighbourhood descriptors, random projections,
sorted-projection distance.  Evaluation only, off the training hot path; an independent implementation of the algorithm of
reference sliced_wasserstein.py:13-133, checked against outputs of that module (``tests/golden/swd_golden.npz``, made by
``tests/golden/make_swd_golden.py``).

Two paths behind the same functions, chosen by the type of the minibatch: numpy arrays are processed on the host (the
reference's arithmetic); ``torch`` tensors on the ROCm device stay there -- pyramid, descriptor gather, standardisation,
projections (one GEMM per repeat) and the sort run as device ops, only the random DRAWS (patch positions, directions) are made
on the host from the same ``RandomState`` in the same order, so both paths see the same patches and directions and differ by
float32 rounding only (``tests/test_metrics_gpu.py`` holds the device path to the reference's golden outputs).

Randomness is explicit: every sampling function takes a ``numpy.random.RandomState``; seeding it like the global
generator the reference uses reproduces the reference's draws."""
from __future__ import annotations

import numpy as np

try:                                # the device path needs torch; the host path must work without it
    import torch
except ImportError:                 # pragma: no cover
    torch = None


def _is_dev(x):
    return torch is not None and isinstance(x, torch.Tensor)


_BINOMIAL = np.array([1.0, 4.0, 6.0, 4.0, 1.0], np.float32) / 16.0      # 5-tap binomial; outer product / 256 = cv2.pyrDown kernel


def _smooth_mirror(x, gain=1.0):
    """Separable 5x5 binomial filter over the last two axes with mirror (reflect-101) borders."""
    if _is_dev(x):
        out = x
        w = torch.tensor(_BINOMIAL, dtype=x.dtype, device=x.device)
        for axis in (2, 3):
            n = out.shape[axis]
            pad = (2, 2, 0, 0) if axis == 3 else (0, 0, 2, 2)
            xp = torch.nn.functional.pad(out, pad, mode="reflect")
            acc = torch.zeros_like(out)
            for j in range(5):                      # the host path's order of accumulation, tap by tap
                acc = acc + w[j] * xp.narrow(axis, j, n)
            out = acc
        return out * gain if gain != 1.0 else out
    out = x
    for axis in (2, 3):
        pad = [(0, 0)] * 4
        pad[axis] = (2, 2)
        xp = np.pad(out, pad, mode="reflect")
        acc = np.zeros_like(out)
        n = out.shape[axis]
        for j, wj in enumerate(_BINOMIAL):
            sl = [slice(None)] * 4
            sl[axis] = slice(j, j + n)
            acc = acc + wj * xp[tuple(sl)]
        out = acc
    return out * np.float32(gain) if gain != 1.0 else out


def pyr_down(minibatch):
    """Gaussian-pyramid step down (NCHW): smooth, keep every second pixel."""
    assert minibatch.ndim == 4
    return _smooth_mirror(minibatch)[:, :, ::2, ::2]


def pyr_up(minibatch):
    """Gaussian-pyramid step up (NCHW): zero-stuff to twice the size, smooth with 4x gain."""
    assert minibatch.ndim == 4
    n, c, h, w = minibatch.shape
    up = minibatch.new_zeros((n, c, 2 * h, 2 * w)) if _is_dev(minibatch) else np.zeros((n, c, 2 * h, 2 * w), minibatch.dtype)
    up[:, :, ::2, ::2] = minibatch
    return _smooth_mirror(up, gain=4.0)


def generate_laplacian_pyramid(minibatch, num_levels):
    """[L0 .. L_{n-1}]: band-pass residuals, last level the low-pass image.  Does not modify its input."""
    levels = [minibatch.to(torch.float32).clone() if _is_dev(minibatch) else np.array(minibatch, dtype=np.float32, copy=True)]
    for _ in range(1, num_levels):
        low = pyr_down(levels[-1])
        levels[-1] = levels[-1] - pyr_up(low)
        levels.append(low)
    return levels


def reconstruct_laplacian_pyramid(pyramid):
    img = pyramid[-1]
    for level in pyramid[-2::-1]:
        img = pyr_up(img) + level
    return img


def get_descriptors_for_minibatch(minibatch, nhood_size, nhoods_per_image, rng):
    """Random nhood_size x nhood_size x 3 patches, nhoods_per_image per image -> [N, 3, n, n]."""
    n_img, chans, height, width = minibatch.shape
    assert chans == 3
    total = nhoods_per_image * n_img
    half = nhood_size // 2
    cx = rng.randint(half, width - half, size=(total, 1, 1, 1))       # drawn in the reference's order: x first, then y
    cy = rng.randint(half, height - half, size=(total, 1, 1, 1))
    img = (np.arange(total) // nhoods_per_image).reshape(total, 1, 1, 1)
    ch = np.arange(3).reshape(1, 3, 1, 1)
    dy = np.arange(-half, half + 1).reshape(1, 1, 1, nhood_size)
    dx = np.arange(-half, half + 1).reshape(1, 1, nhood_size, 1)
    if _is_dev(minibatch):                          # the same index arrays, uploaded: one gather on the device
        idx = [torch.from_numpy(np.ascontiguousarray(np.broadcast_to(a, (total, 3, nhood_size, nhood_size)))).to(minibatch.device)
               for a in (img, ch, cy + dy, cx + dx)]
        return minibatch[idx[0], idx[1], idx[2], idx[3]]
    return minibatch[img, ch, cy + dy, cx + dx]


def finalize_descriptors(desc):
    """Concatenate, standardise per channel, flatten to [N, 3*n*n]."""
    if isinstance(desc, list):
        desc = torch.cat(desc, dim=0) if _is_dev(desc[0]) else np.concatenate(desc, axis=0)
    assert desc.ndim == 4
    if _is_dev(desc):
        desc = desc - desc.mean(dim=(0, 2, 3), keepdim=True)
        desc = desc / desc.std(dim=(0, 2, 3), keepdim=True, unbiased=False)      # numpy's std: population
        return desc.reshape(desc.shape[0], -1)
    desc = desc - desc.mean(axis=(0, 2, 3), keepdims=True)
    desc = desc / desc.std(axis=(0, 2, 3), keepdims=True)
    return desc.reshape(desc.shape[0], -1)


def sliced_wasserstein(A, B, dir_repeats, dirs_per_repeat, rng):
    """Mean |sorted projection difference| over random unit directions."""
    assert A.ndim == 2 and A.shape == B.shape
    per_repeat = []
    for _ in range(dir_repeats):
        dirs = rng.randn(A.shape[1], dirs_per_repeat)
        dirs = (dirs / np.sqrt((dirs ** 2).sum(axis=0, keepdims=True))).astype(np.float32)
        if _is_dev(A):
            d = torch.from_numpy(dirs).to(A.device)
            pa = torch.sort(A @ d, dim=0).values
            pb = torch.sort(B @ d, dim=0).values
            per_repeat.append(float((pa - pb).abs().mean()))
            continue
        pa = np.sort(A @ dirs, axis=0)
        pb = np.sort(B @ dirs, axis=0)
        per_repeat.append(np.abs(pa - pb).mean())
    return float(np.mean(per_repeat))


class API:
    """Same driver protocol as the reference's ``API`` (begin / feed / end), RNG explicit."""

    def __init__(self, image_shape, seed=None):
        self.nhood_size, self.nhoods_per_image, self.dir_repeats, self.dirs_per_repeat = 7, 128, 4, 128
        self.resolutions = []
        res = image_shape[1]
        while res >= 16:
            self.resolutions.append(res)
            res //= 2
        self.rng = np.random.RandomState(seed)

    def get_metric_names(self):
        return ["SWDx1e3_%d" % r for r in self.resolutions] + ["SWDx1e3_avg"]

    def begin(self, mode):
        assert mode in ("warmup", "reals", "fakes")
        self.descriptors = [[] for _ in self.resolutions]

    def feed(self, mode, minibatch):
        for lod, level in enumerate(generate_laplacian_pyramid(minibatch, len(self.resolutions))):
            self.descriptors[lod].append(get_descriptors_for_minibatch(level, self.nhood_size, self.nhoods_per_image, self.rng))

    def end(self, mode):
        desc = [finalize_descriptors(d) for d in self.descriptors]
        del self.descriptors
        if mode in ("warmup", "reals"):
            self.desc_real = desc
        dist = [sliced_wasserstein(r, f, self.dir_repeats, self.dirs_per_repeat, self.rng) * 1e3 for r, f in zip(self.desc_real, desc)]
        return dist + [float(np.mean(dist))]
