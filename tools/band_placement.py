#!/usr/bin/env python
"""Where and when the waves of blur_band_t_kernel ran, from a -DBG_DIAG -DBLUR_BAND_STAMP build
(tools/build_variant.sh band_stamp blur.hip -DBG_DIAG -DBLUR_BAND_STAMP; BGAN_HIP_LIB=tools/_build/libbgan_band_stamp.so).
Prints waves per (XCC, SE, CU, SIMD) slot, the span of the launch and the start times of the workgroups.
Usage: band_placement.py B H W C sigma"""
import collections
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blurred_gan_amd import _lib, ops  # noqa: E402

B, H, W, C = (int(v) for v in sys.argv[1:5])
sigma = float(sys.argv[5])
x = torch.rand(B, H, W, C, device="cuda") * 2 - 1
y = torch.empty_like(x)
ks, se, nt = ops.blur_policy(sigma, H, W)
taps = torch.tensor(ops.gauss_kernel_1d(se, ks), device="cuda")
tmp = torch.empty(ops.blur_workspace_bytes(B, H, W, C, nt) // 4 + 4, device="cuda")
for _ in range(3):
    ops.blur_nhwc(x, y, taps, nt, tmp)
torch.cuda.synchronize()
ops.prof_reset(); ops.prof_enable(True)
for _ in range(10):
    ops.blur_nhwc(x, y, taps, nt, tmp)
torch.cuda.synchronize()
recs = ops.prof_records(); ops.prof_enable(False); ops.prof_reset()
pass_us = sum(r[1] for r in recs) / len(recs) * 1e3
lib = _lib.load()
NW = 3 if C == 3 else 4
nwg = B * ((H + 127) // 128) * ((W * C + NW * 32 - 1) // (NW * 32))
n = min(4096, nwg * NW)
buf = np.zeros(4096 * 4, dtype=np.uint64)
fn = lib.bg_dbg_band_read
fn.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
rc = fn(buf.ctypes.data, buf.size)
assert rc == 0, rc
st = buf.reshape(4096, 4)[:n]
hw = st[:, 3] & 0xFFFFFFFF
xcc = ((st[:, 3] >> 32) & 0xF).astype(np.int64)
raw = [st[:, k].astype(np.int64) for k in range(3)]
se_k, sh_k, cu_k = (hw >> 13) & 7, (hw >> 12) & 1, (hw >> 8) & 15
cukey = ((xcc * 8 + se_k.astype(np.int64)) * 2 + sh_k.astype(np.int64)) * 16 + cu_k.astype(np.int64)
t0 = np.zeros(n, dtype=np.int64)                # the counters are not synchronised across the chip: times relative to the CU's first wave
for key in np.unique(cukey):
    t0[cukey == key] = raw[0][cukey == key].min()
print("distinct XCC ids read:", np.unique(xcc).tolist())
start, mid, end = (r - t0 for r in raw)
simd, cu, sh, se_ = (hw >> 4) & 3, (hw >> 8) & 15, (hw >> 12) & 1, (hw >> 13) & 7
print(f"{n} waves of {nwg} workgroups ({NW} waves each), taps {nt}")
print(f"launch span {end.max()} ticks; wave run time mean {np.mean(end - start):.0f}, "
      f"main loop mean {np.mean(mid - start):.0f}, store mean {np.mean(end - mid):.0f}")
slots = collections.Counter(zip(xcc.tolist(), se_.tolist(), sh.tolist(), cu.tolist(), simd.tolist()))
cus = collections.Counter(zip(xcc.tolist(), se_.tolist(), sh.tolist(), cu.tolist()))
print(f"distinct CUs {len(cus)}, waves per CU: {sorted(collections.Counter(cus.values()).items())}")
print(f"distinct SIMDs {len(slots)}, waves per SIMD: {sorted(collections.Counter(slots.values()).items())}")
per_simd = collections.Counter(simd.tolist())
print("waves by SIMD index:", sorted(per_simd.items()))
late = start > np.median(end - start) * 0.5
print(f"waves that started after half a wave run time (per-XCD clock): {int(late.sum())} ({100.0 * late.mean():.1f} %)")
hist, edges = np.histogram(start, bins=10)
print("start-time histogram (ticks):", list(zip(edges[:-1].round(0).tolist(), hist.tolist())))
hist, edges = np.histogram(end, bins=10)
print("end-time histogram (ticks):", list(zip(edges[:-1].round(0).tolist(), hist.tolist())))
spans = np.array([end[cukey == key].max() for key in np.unique(cukey)])
print(f"mean pass {pass_us:.1f} us by events -> {spans.mean() / pass_us:.0f} ticks per us if the waves filled the pass")
print(f"per-CU span (first start .. last end): mean {spans.mean():.0f} ticks, max {spans.max()}, against a mean wave run time of {np.mean(end - start):.0f}")
