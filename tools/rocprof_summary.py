#!/usr/bin/env python
"""Turns a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the small summary committed
under profiles/ (per-kernel calls / total / average duration).  Usage: rocprof_summary.py <dir> <out.md> [title]"""
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out)
    f = glob.glob(os.path.join(d, "**", "*_kernel_stats.csv"), recursive=True)
    assert f, f"no *_kernel_stats.csv under {d}"
    rows = list(csv.DictReader(open(f[0])))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats --output-format csv` ({os.path.basename(f[0])}); "
                 f"total kernel time {total / 1e6:.3f} ms\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            if float(r["TotalDurationNs"]) / total < 0.0005:
                continue
            fh.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    print(out)


if __name__ == "__main__":
    main()
