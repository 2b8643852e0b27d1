#!/usr/bin/env python
"""Runs N applications of bg_blur_nhwc_f32 on one shape / sigma and nothing else (the workload for rocprofv3 passes).
Usage: blur_run.py B H W C sigma [applications]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from blurred_gan_amd import ops  # noqa: E402

B, H, W, C = (int(v) for v in sys.argv[1:5])
sigma = float(sys.argv[5])
n = int(sys.argv[6]) if len(sys.argv) > 6 else 20
x = torch.rand(B, H, W, C, device="cuda") * 2 - 1
y = torch.empty_like(x)
ks, se, nt = ops.blur_policy(sigma, H, W)
taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
nb = ops.blur_workspace_bytes(B, H, W, C, nt)
tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None
for i in range(n):
    ops.blur_nhwc(x if i % 2 == 0 else y, y if i % 2 == 0 else x, taps, nt, tmp)
torch.cuda.synchronize()
print("done", (B, H, W, C), nt, "taps", n, "applications")
