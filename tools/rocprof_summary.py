#!/usr/bin/env python
"""Turns a rocprofv3 `--kernel-trace --stats --output-format csv` directory into the small summary committed
under profiles/ (per-kernel calls / total / average duration).  Usage: rocprof_summary.py <dir> <out.md> [title]

The directory must hold exactly ONE *_kernel_stats.csv: a directory reused across rounds once made this tool summarise a
stale run under a fresh title.  The summary names the CSV (path relative to the repository) and its mtime."""
import csv
import glob
import os
import sys
import time


def find_one(d, pattern):
    f = sorted(glob.glob(os.path.join(d, "**", pattern), recursive=True))
    assert f, f"no {pattern} under {d}"
    assert len(f) == 1, f"{len(f)} files match {pattern} under {d} (profile into a fresh directory): {f}"
    return f[0]


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[3] if len(sys.argv) > 3 else os.path.basename(out)
    f = find_one(d, "*_kernel_stats.csv")
    rows = list(csv.DictReader(open(f)))
    total = sum(float(r["TotalDurationNs"]) for r in rows)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    rel = os.path.relpath(os.path.abspath(f), root)
    stamp = time.strftime("%Y-%m-%d %H:%M:%S", time.gmtime(os.path.getmtime(f)))
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\nsource: `rocprofv3 --kernel-trace --stats --output-format csv`, `{rel}` (written {stamp} UTC); "
                 f"total kernel time {total / 1e6:.3f} ms\n\n| kernel | calls | total ms | avg us | min us | max us | % |\n|---|---|---|---|---|---|---|\n")
        for r in rows:
            if float(r["TotalDurationNs"]) / total < 0.0005:
                continue
            fh.write(f"| `{r['Name'][:110]}` | {r['Calls']} | {float(r['TotalDurationNs']) / 1e6:.3f} | {float(r['AverageNs']) / 1e3:.2f} | "
                     f"{float(r['MinNs']) / 1e3:.2f} | {float(r['MaxNs']) / 1e3:.2f} | {float(r['Percentage']):.2f} |\n")
    print(out)


if __name__ == "__main__":
    main()
