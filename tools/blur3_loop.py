"""The critic batch's blur of one discriminator_step (fakes, reals, x-hat), fused against separate launches, timed with events.
Usage: python tools/blur3_loop.py B H W C sigma [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops

B, H, W, C = map(int, sys.argv[1:5])
sigma = float(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 100
ks, se, nt = ops.blur_policy(sigma, H, W)
taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
f = torch.rand(B, H, W, C, device="cuda") * 2 - 1
r = torch.rand(B, H, W, C, device="cuda") * 2 - 1
a = torch.rand(B, device="cuda")
y3 = torch.empty(3 * B, H, W, C, device="cuda")
xh = torch.empty_like(f)


def fused():
    ops.blur3_lerp(f, r, a, y3, taps, nt)


def separate():
    ops.lerp(r, f, a, xh)
    for i, src in enumerate((f, r, xh)):
        ops.blur_nhwc(src, y3[i * B:(i + 1) * B], taps, nt, None)


for name, fn in (("fused (1 launch)", fused), ("lerp + 3 blurs", separate), ("fused (1 launch)", fused), ("lerp + 3 blurs", separate)):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / iters * 1e3
    byt = 5.0 * B * H * W * C * 4          # algorithmic: read fakes and reals, write three blurred slices
    print(f"{name:<18} {nt} taps: {us:7.2f} us  = {byt / us / 1e3:7.1f} GB/s algorithmic ({byt / us / 1e3 / 8000:.3f} of HBM)")
