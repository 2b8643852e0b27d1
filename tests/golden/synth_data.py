"""Seeded synthetic image distribution for the statistical curve-parity run (no dataset is reachable offline): soft blobs on a dark
background, 1-channel or RGB, values in [-1, 1] like the demos' `(x - 127.5) / 127.5` inputs (demo_mnist.py:24-31).  Pure numpy,
so the golden maker (CPU, build container) and the GPU test draw the very same images from a seed."""
import numpy as np


def blob_dataset(n, size=28, channels=1, seed=0):
    """[n, size, size, channels] float32: 1-3 Gaussian blobs per image (random centre, width 1.5 ... size/5 pixels, amplitude
    0.5 ... 1), summed, clipped to [0, 1] and mapped to [-1, 1]."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:size, 0:size].astype(np.float32)
    out = np.empty((n, size, size, channels), np.float32)
    for i in range(n):
        img = np.zeros((size, size), np.float32)
        for _ in range(int(rng.integers(1, 4))):
            cy, cx = rng.uniform(0.2 * size, 0.8 * size, 2)
            s = rng.uniform(1.5, size / 5.0)
            img += np.float32(rng.uniform(0.5, 1.0)) * np.exp(-((yy - cy) ** 2 + (xx - cx) ** 2) / np.float32(2 * s * s))
        img = np.clip(img, 0.0, 1.0) * 2.0 - 1.0
        for c in range(channels):
            out[i, :, :, c] = img * np.float32(1.0 - 0.15 * c)
    return out


def batches(data, batch, steps, seed):
    """`steps` minibatches of `batch` images: fresh permutation per epoch, seeded."""
    rng = np.random.default_rng(seed)
    n = len(data)
    order = rng.permutation(n)
    pos = 0
    for _ in range(steps):
        if pos + batch > n:
            order, pos = rng.permutation(n), 0
        yield data[order[pos:pos + batch]]
        pos += batch


def sigma_schedule(step, start=2.0, rate=0.99, floor=0.05):
    """The blur sigma of training step `step`: exponential decay like BlurDecayController's (callbacks.py:51-57) but fast enough
    that the taps change DURING a 400-step run (13 taps at step 0, 3 taps from step ~140 on)."""
    return max(float(start) * float(rate) ** int(step), float(floor))


def to_swd_input(x):
    """NHWC in [-1, 1] -> the SWD code's NCHW 0..255 minibatch with 3 channels (demo_mnist.py:180-184)."""
    x = np.asarray(x, np.float32)
    x = (x + 1.0) * 0.5
    if x.shape[-1] == 1:
        x = np.repeat(x, 3, axis=-1)
    return np.ascontiguousarray(x.transpose(0, 3, 1, 2)) * 255.0
