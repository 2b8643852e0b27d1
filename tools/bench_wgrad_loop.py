"""Back-to-back launches of one layer's filter gradient (long enough for counters and a settled clock).
Usage: python tools/bench_wgrad_loop.py B H W Ci Co [iters]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from blurred_gan_amd import ops

B, H, W, Ci, Co = map(int, sys.argv[1:6])
iters = int(sys.argv[6]) if len(sys.argv) > 6 else 30
x = torch.rand(B, H, W, Ci, device="cuda") - 0.5
dy = torch.rand(B, H // 2, W // 2, Co, device="cuda") - 0.5
dw = torch.empty(5, 5, Ci, Co, device="cuda")
nb = ops.conv2d_bwd_filter_workspace_bytes(B, H, W, Ci, Co, 5, 2)
ws = torch.empty(nb // 4 + 4, device="cuda") if nb else None
for _ in range(iters):
    ops.conv2d_bwd_filter(x, dy, dw, 5, 2, 0.0, 1.0, ws)
torch.cuda.synchronize()
