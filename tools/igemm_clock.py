#!/usr/bin/env python
"""Shader clock the gather-GEMM really runs at: a -DBG_DIAG -DIGEMM_CLOCK build of conv_igemm.hip leaves per-workgroup (s_memtime,
s_memrealtime) tick counts of the K loop; the ratio x 100 MHz is the clock (cdna_hip_programming.md section 7).
  BGAN_HIP_LIB=tools/_build/libbgan_igemm_CLOCK.so python tools/igemm_clock.py [--arch celeba64] [--batch 256] [--only G3,G4]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402
from blurred_gan_amd import _lib, ops  # noqa: E402
from bench_conv import LAYERS  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--arch", default="celeba64")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--only", default="")
    ap.add_argument("--reps", type=int, default=30)
    ap.add_argument("--dump", action="store_true", help="per-XCD statistics of the last launch's workgroup records")
    a = ap.parse_args()
    lib = _lib.load()
    dump = lib.bg_diag_clock_dump
    dump.restype = C.c_int
    dump.argtypes = [C.POINTER(C.c_uint64), C.c_int]
    B = a.batch
    print(f"{'layer':<20}{'op':<7}{'us/launch':>10}{'loop us':>9}{'clock GHz':>10}{'% of 157.3 at that clock':>26}   workgroups: kernel span / spread of the entries / mean prologue / mean K loop (us) | per CU")
    for name, H, W, Ci, Co, s in LAYERS[a.arch]:
        if a.only and not any(o in name for o in a.only.split(",")):
            continue
        Ho, Wo = -(-H // s), -(-W // s)
        x = torch.rand(B, H, W, Ci, device="cuda") - 0.5
        dy = torch.rand(B, Ho, Wo, Co, device="cuda") - 0.5
        w = torch.rand(5, 5, Ci, Co, device="cuda") - 0.5
        wT = ops.transpose_last2(w, torch.empty(w.numel(), device="cuda"), 25, Ci, Co)
        y, dx = torch.empty_like(dy), torch.empty_like(x)
        nf, nd = ops.conv2d_splitk_workspace_bytes(False, B, H, W, Ci, Co, 5, s), ops.conv2d_splitk_workspace_bytes(True, B, H, W, Ci, Co, 5, s)
        wsk = torch.empty(max(nf, nd) // 4 + 4, device="cuda")
        ef, ed = ops.epilogue(ws=wsk if nf else None), ops.epilogue(ws=wsk if nd else None)
        fl = 2.0 * B * Ho * Wo * Ci * Co * 25
        for op, fn in (("fwd", lambda: ops.conv2d_fwd(x, wT, y, 5, s, ef)), ("dgrad", lambda: ops.conv2d_bwd_data(dy, w, dx, 5, s, ed))):
            for _ in range(3):
                fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(a.reps):
                fn()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) * 1e3 / a.reps
            buf = (C.c_uint64 * (6 * 8192))()
            nd_ = dump(buf, 8192)
            if nd_ <= 0:
                print(f"{name:<20}{op:<7}{us:10.1f}   (no igemm launch / no stamps: {nd_})")
                continue
            r = np.frombuffer(buf, dtype=np.uint64)[:6 * nd_].reshape(nd_, 6).astype(np.int64)
            t0 = r[:, 1].min()
            loop = (r[:, 3] - r[:, 2]) / 100.0
            pro = (r[:, 2] - r[:, 1]) / 100.0
            end = (r[:, 3] - t0) / 100.0
            ghz = float(r[:, 0].sum() / (r[:, 3] - r[:, 2]).sum() * 0.1)
            tf = fl / (us * 1e-6) / 1e12
            xcc = r[:, 5] & 15
            cu = xcc * 128 + ((r[:, 4] >> 13) & 7) * 32 + ((r[:, 4] >> 12) & 1) * 16 + ((r[:, 4] >> 8) & 15)      # HW_ID: se_id [15:13], sh_id [12], cu_id [11:8]
            cus = np.unique(cu)
            last = np.array([end[cu == c].max() for c in cus])
            first = np.array([end[cu == c].min() for c in cus])
            print(f"{name:<20}{op:<7}{us:10.1f}{loop.mean():9.1f}{ghz:10.3f}{100 * tf / (157.3 * ghz / 2.4):26.1f}   {nd_}: {end.max():.1f} / {(r[:, 1].max() - t0) / 100.0:.1f} / "
                  f"{pro.mean():.1f} / {loop.mean():.1f} | CUs {len(cus)}: last end {last.min():.1f}..{last.max():.1f}, first workgroup to end on a CU {first.mean():.1f} "
                  f"(wgs per CU {min((cu == c).sum() for c in cus)}..{max((cu == c).sum() for c in cus)})")
            if a.dump:
                print("    end-time percentiles (us) 5/25/50/75/95/100: " + " ".join(f"{np.percentile(end, q):.1f}" for q in (5, 25, 50, 75, 95, 100)))
                for x in range(8):
                    m = xcc == x
                    if m.any():
                        clk = r[m, 0].sum() / (r[m, 3] - r[m, 2]).sum() * 0.1
                        print(f"    xcc {x}: wgs {m.sum():4d}  loop mean {loop[m].mean():7.1f}  last end {end[m].max():7.1f}  clock {clk:.3f} GHz")


if __name__ == "__main__":
    main()
