"""Error of the conv forward / data gradient against an fp64 reference, for whichever arithmetic BGAN_CONV_MATH selects.
Run twice:  python tests/x6_accuracy.py   and   BGAN_CONV_MATH=bf16x6 python tests/x6_accuracy.py"""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from blurred_gan_amd import ops
from oracle import np_ops as O

print("BGAN_CONV_MATH =", os.environ.get("BGAN_CONV_MATH", "(fp32 MFMA)"))
for (B, H, W, Ci, Co, s) in [(8, 32, 32, 64, 128, 2), (8, 16, 16, 128, 256, 2), (16, 8, 8, 256, 512, 2), (4, 16, 16, 256, 128, 1)]:
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(B, H, W, Ci))
    w = rng.uniform(-1, 1, size=(5, 5, Ci, Co)) / np.sqrt(25 * Ci)
    Ho, Wo = -(-H // s), -(-W // s)
    dy = rng.uniform(-1, 1, size=(B, Ho, Wo, Co))
    dev = lambda a: torch.from_numpy(np.ascontiguousarray(a, dtype=np.float32)).cuda()
    ref_y = O.conv2d_fwd(x.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64), s)
    ref_dx = O.conv2d_bwd_data(dy.astype(np.float32).astype(np.float64), w.astype(np.float32).astype(np.float64), s, (H, W))
    wT = dev(np.transpose(w, (0, 1, 3, 2)))
    y = ops.conv2d_fwd(dev(x), wT, torch.empty(ref_y.shape, device="cuda"), 5, s).cpu().numpy().astype(np.float64)
    dx = ops.conv2d_bwd_data(dev(dy), dev(w), torch.empty(x.shape, device="cuda"), 5, s).cpu().numpy().astype(np.float64)
    for nm, got, ref in (("fwd  ", y, ref_y), ("dgrad", dx, ref_dx)):
        err = np.abs(got - ref)
        print(f"{(B,H,W,Ci,Co,s)} {nm} K={25*(Ci if nm=='fwd  ' else Co):5d}  max|err| {err.max():.3e}  rms err {np.sqrt((err**2).mean()):.3e}  rms ref {np.sqrt((ref**2).mean()):.3e}"
              f"  max rel-to-rms {err.max()/np.sqrt((ref**2).mean()):.3e}")
