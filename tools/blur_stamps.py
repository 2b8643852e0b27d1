#!/usr/bin/env python
"""Phase timeline of blur_strip_kernel from a -DBG_DIAG -DBLUR_STRIP_STAMP build (tools/build_variant.sh stamp blur.hip -DBG_DIAG -DBLUR_STRIP_STAMP;
BGAN_HIP_LIB=tools/_build/libbgan_stamp.so): mean cycles per phase and iteration over all workgroups.
Usage: blur_stamps.py B H W C sigma"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
from blurred_gan_amd import ops  # noqa: E402

B, H, W, C = (int(v) for v in sys.argv[1:5])
sigma = float(sys.argv[5])
print("==", os.environ.get("BGAN_HIP_LIB", "in-tree library"), (B, H, W, C), "sigma", sigma)
x = torch.rand(B, H, W, C, device="cuda") * 2 - 1
y = torch.empty_like(x)
ks, se, nt = ops.blur_policy(sigma, H, W)
taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
nwg = 8 * ((B + 7) // 8) * ((W + 31) // 32)
dbg = torch.zeros(nwg * 12 * 8 * 2 + 16, dtype=torch.float32, device="cuda")      # 8-byte stamps
for _ in range(5):
    ops.blur_nhwc(x, y, taps, nt, dbg)
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
for _ in range(10):
    ops.blur_nhwc(x, y, taps, nt, dbg)
torch.cuda.synchronize()
recs = ops.prof_records(); ops.prof_enable(False); ops.prof_reset()
ms = sum(r[1] for r in recs) / 10
print(f"kernel time {ms * 1e3:.1f} us = {8.0 * B * H * W * C / ms / 1e6:.0f} GB/s algorithmic")
if "stamp" not in os.environ.get("BGAN_HIP_LIB", ""):
    sys.exit(0)
dbg.zero_()
ops.blur_nhwc(x, y, taps, nt, dbg)
torch.cuda.synchronize()
st = dbg[:nwg * 12 * 8 * 2].view(torch.int64).view(nwg, 12, 8).cpu().double()
names = ["W phase", "barrier", "H phase+mem", "barrier"]
n_iter = int((st[0, :, 0] > 0).sum())
t0 = st[:, 0, 0].min()
print(f"{nwg} workgroups, {n_iter} iterations; first start .. last end = {(st[:, n_iter - 1, 4].max() - t0):.0f} ticks; "
      f"workgroup start spread {(st[:, 0, 0].max() - t0):.0f}, mean workgroup span {(st[:, n_iter - 1, 4] - st[:, 0, 0]).mean():.0f}")
print("iter " + " ".join(f"{n[:12]:>13}" for n in names) + "         total")
for j in range(n_iter):
    d = [(st[:, j, k + 1] - st[:, j, k]).mean().item() for k in range(4)]
    print(f"{j:4d} " + " ".join(f"{v:13.0f}" for v in d) + f" {sum(d):13.0f}")
