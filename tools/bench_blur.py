#!/usr/bin/env python
"""C5 blur stress (BASELINE.json configs[4], per-rank share): 64 x 256x256x3 images, sigma 5 / 23.5 / 42.33 -> 31 / 143 / 255
taps, plus the C2 and C4 image sizes.  Reports ms, algorithmic GB/s (8*H*W*C bytes per image) against the 8 TB/s HBM
peak and direct-form GFLOP/s (4*T*H*W*C flop per image) -- the large-tap cases are VALU-bound, not HBM-bound."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from blurred_gan_amd import ops  # noqa: E402


def main():
    print(f"{'shape':<22}{'sigma':>7}{'taps':>6}{'ms':>9}{'GB/s(alg)':>11}{'%HBM':>7}{'TFLOP/s':>9}  kernels")
    cases = (((256, 64, 64, 3), (5.0,)), ((128, 128, 128, 3), (5.0,)), ((64, 256, 256, 3), (5.0, 23.5, 42.34)))
    if "--extra" in sys.argv:       # the band passes at other channel counts, a ragged width and a width that is not float4-addressable
        cases += (((64, 256, 256, 1), (23.5,)), ((64, 256, 256, 4), (5.0, 23.5)), ((64, 256, 256, 2), (23.5,)), ((64, 250, 250, 3), (23.5,)),
                  ((16, 512, 512, 3), (23.5,)))
    for (B, H, W, C), sigmas in cases:
        x = torch.rand(B, H, W, C, device="cuda") * 2 - 1
        y = torch.empty_like(x)
        for sg in sigmas:
            ks, se, nt = ops.blur_policy(sg, H, W)
            taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
            nb = ops.blur_workspace_bytes(B, H, W, C, nt)
            tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None
            for _ in range(2):
                ops.blur_nhwc(x, y, taps, nt, tmp)
            torch.cuda.synchronize()
            ops.prof_reset(); ops.prof_enable(True)
            for _ in range(10):
                ops.blur_nhwc(x, y, taps, nt, tmp)
            torch.cuda.synchronize()
            recs = ops.prof_records(); ops.prof_enable(False); ops.prof_reset()
            ms = sum(r[1] for r in recs) / 10
            gbs = 8.0 * B * H * W * C / (ms * 1e-3) / 1e9
            tf = 4.0 * nt * B * H * W * C / (ms * 1e-3) / 1e12
            print(f"{str((B, H, W, C)):<22}{sg:7.2f}{nt:6d}{ms:9.4f}{gbs:11.1f}{100 * gbs / 8000:7.1f}{tf:9.2f}  {','.join(sorted({r[0] for r in recs}))}")


if __name__ == "__main__":
    main()
