"""Back-to-back blur launches of one configuration, timed with events.  Usage: python tools/blur_loop.py B H W C sigma [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops

B, H, W, C = map(int, sys.argv[1:5])
sigma = float(sys.argv[5])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 50
ks, se, nt = ops.blur_policy(sigma, H, W)
taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
x = torch.rand(B, H, W, C, device="cuda") * 2 - 1
y = torch.empty_like(x)
nb = ops.blur_workspace_bytes(B, H, W, C, nt)
tmp = torch.empty(nb // 4 + 4, device="cuda") if nb else None
for _ in range(5):
    ops.blur_nhwc(x, y, taps, nt, tmp)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(iters):
    ops.blur_nhwc(x, y, taps, nt, tmp)
e1.record()
torch.cuda.synchronize()
print(f"{os.environ.get('BGAN_HIP_LIB', 'default').split('/')[-1]:<28} {nt} taps: {e0.elapsed_time(e1) / iters * 1e3:.2f} us per application")
