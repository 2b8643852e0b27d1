#!/usr/bin/env python
"""Per-kernel matrix-pipe utilisation out of one rocprofv3 --pmc pass (tools/pmc_mfma.sh):
  MFMA busy      = SQ_VALU_MFMA_BUSY_CYCLES / (4 x SQ_BUSY_CU_CYCLES): share of the CU-busy time in which a SIMD's matrix pipe works
                   (4 SIMDs per CU; 1.0 = every SIMD of every busy CU issues MFMAs back to back)
  GUI-active GHz = GRBM_GUI_ACTIVE / 8 / duration (the counter is the sum over the 8 XCDs).  NOT the shader clock: the GUI is active
                   before and after the waves run, so the ratio reads 3-7 "GHz" on kernels of a few microseconds (round-4 verdict) and
                   2.41 on every long one, while in-kernel stamps (s_memtime / s_memrealtime, tools/igemm_clock.py,
                   profiles/r05_a_gather_gemm_limits.md) put the same gather-GEMM launches at 1.9-2.37 GHz.  It is printed for
                   kernels of >= 30 us only, as an upper bound.
  of peak        = MFMA busy x min(GUI-active GHz, 2.4) / 2.4 GHz, kernels of >= 30 us only: an UPPER bound on the fraction of the
                   datasheet matrix rate the pipe delivered (the busy share itself does not depend on the clock).
Usage: pmc_mfma_report.py <dir> <out.md> [title]"""
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out)
    cf = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    kf = glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True)
    assert len(cf) == 1 and len(kf) == 1, (cf, kf)
    dur = {}
    for r in csv.DictReader(open(kf[0])):
        dur[int(r["Dispatch_Id"])] = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) * 1e-9
    per = {}
    for r in csv.DictReader(open(cf[0])):
        k = int(r["Dispatch_Id"])
        e = per.setdefault(k, {"name": r["Kernel_Name"]})
        e[r["Counter_Name"]] = e.get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
    ids = sorted(per)
    ids = ids[len(ids) // 3:]                                # steady state: the first third is warm-up / recording steps
    agg = {}
    for k in ids:
        e = per[k]
        if k not in dur or "SQ_BUSY_CU_CYCLES" not in e:
            continue
        name = e["name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
        a = agg.setdefault(name, [0, 0.0, 0.0, 0.0, 0.0])
        a[0] += 1
        a[1] += dur[k]
        a[2] += e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
        a[3] += e.get("SQ_BUSY_CU_CYCLES", 0.0)
        a[4] += e.get("GRBM_GUI_ACTIVE", 0.0)
    rows = sorted(agg.items(), key=lambda kv: -kv[1][1])
    total = sum(v[1] for _, v in rows)
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\nsource: `{os.path.relpath(cf[0])}` (`rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES GRBM_GUI_ACTIVE "
                 f"--kernel-trace`), last two thirds of the dispatches; durations from the same pass's kernel trace\n\n"
                 "| kernel | launches | total ms | avg us | MFMA busy / (4 x CU busy) | GUI-active GHz (>= 30 us kernels; upper bound of the clock) | of the 2.4 GHz matrix peak (upper bound) |\n|---|---|---|---|---|---|---|\n")
        for name, (n, t, mf, cu, gui) in rows:
            if t / total < 0.002:
                continue
            busy = mf / (4.0 * cu) if cu else 0.0
            clk = gui / 8.0 / t / 1e9 if t else 0.0
            if t / n >= 30e-6:
                fh.write(f"| `{name[:100]}` | {n} | {t * 1e3:.3f} | {t / n * 1e6:.1f} | {busy:.3f} | {clk:.2f} | {busy * min(clk, 2.4) / 2.4:.3f} |\n")
            else:                                               # GUI-active over the duration says nothing about a kernel this short
                fh.write(f"| `{name[:100]}` | {n} | {t * 1e3:.3f} | {t / n * 1e6:.1f} | {busy:.3f} | - | - |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
