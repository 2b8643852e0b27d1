"""Blur kernel timing sweep: python tools/blur_sweep.py  (per-application time by kernel family for several sizes / sigmas)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops
for shape in [(128, 128, 128, 3), (64, 96, 96, 3), (64, 256, 256, 3), (128, 128, 128, 1)]:
    for std in [0.7, 1.5, 2.5, 3.4, 5.0, 23.5]:
        B, H, W, C = shape
        ks, se, nt = ops.blur_policy(std, H, W)
        taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
        x = torch.rand(*shape, device="cuda"); y = torch.empty_like(x); tmp = torch.empty_like(x)
        for _ in range(3): ops.blur_nhwc(x, y, taps, nt, tmp)
        ops.prof_reset(); ops.prof_enable(True)
        for _ in range(5): ops.blur_nhwc(x, y, taps, nt, tmp)
        torch.cuda.synchronize()
        recs = ops.prof_records(); ops.prof_enable(False)
        d = {}
        for n, ms, fl, by in recs: d[n] = d.get(n, 0.0) + ms / 5
        tot = sum(d.values())
        print(shape, "taps", nt, "total %.1f us" % (tot * 1e3), {n: round(v * 1e3, 1) for n, v in d.items()}, "%.2f TB/s" % (8.0 * x.numel() / tot / 1e9))
