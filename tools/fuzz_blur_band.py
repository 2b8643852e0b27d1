#!/usr/bin/env python
"""Seeded fuzz of the band passes of the blur (more than 65 taps, or 2 / 4 channels at any tap count) against the oracle: image
sizes up to 400 pixels with ragged column groups, partial row blocks and widths that are or are not float4-addressable.
Usage: fuzz_blur_band.py [cases] [seed]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import test_blur_gpu as T  # noqa: E402
from oracle import np_ops as O  # noqa: E402

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 777)
bad = n = 0
while n < cases:
    B, H, W = int(rng.integers(1, 4)), int(rng.integers(65, 400)), int(rng.integers(65, 400))
    C = int(rng.choice([1, 2, 3, 3, 3, 4]))
    if B * H * W * C > 600_000:
        continue
    std = float(rng.choice([9.0, 12.0, 15.0, 23.5, 33.0, 42.34])) if C in (1, 3) else float(rng.choice([1.0, 3.0, 5.0, 12.0, 23.5]))
    x = rng.uniform(-1, 1, size=(B, H, W, C)).astype(np.float32)
    n += 1
    try:
        y, (_, _, nt) = T._run(x, std)
        ref = O.blur_images(x.astype(np.float64), std)
        np.testing.assert_allclose(y, ref, rtol=T.POINT_RTOL, atol=T.POINT_ATOL * 4)
    except Exception as e:          # noqa: BLE001
        bad += 1
        print("FAIL", (B, H, W, C), std, str(e)[:200].replace("\n", " "), flush=True)
    if n % 20 == 0:
        print(n, "cases", bad, "failures", flush=True)
print("done", n, "cases,", bad, "failures")
sys.exit(1 if bad else 0)
