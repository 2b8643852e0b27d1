#!/usr/bin/env python
"""GPU-side timeline of the training step out of a `rocprofv3 --kernel-trace --output-format csv` directory: how much of the
timed region the card spends INSIDE kernels and how much BETWEEN them (launch gaps), per step.
Usage: step_timeline.py <dir> <out.md> <launches_per_step_hint or 0> [title]
The last 60 % of the trace (steady state: warm-up, recording and allocation steps are at the front) is analysed."""
import csv
import glob
import os
import sys


def main():
    d, out = sys.argv[1], sys.argv[2]
    title = sys.argv[4] if len(sys.argv) > 4 else os.path.basename(out)
    fs = sorted(glob.glob(os.path.join(d, "**", "*_kernel_trace.csv"), recursive=True))
    assert len(fs) == 1, fs
    rows = list(csv.DictReader(open(fs[0])))
    ev = sorted((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in rows)
    ev = ev[int(len(ev) * 0.4):]
    span = ev[-1][1] - ev[0][0]
    busy = sum(e - s for s, e, _ in ev)
    gaps = [ev[i + 1][0] - ev[i][1] for i in range(len(ev) - 1)]
    pos = [g for g in gaps if g > 0]
    # step boundaries: the metric read-back separates steps -- a gap much longer than the median
    med = sorted(pos)[len(pos) // 2] if pos else 0
    big = [g for g in pos if g > 10 * max(med, 1000)]
    small = [g for g in pos if g <= 10 * max(med, 1000)]
    hist = {}
    for g in small:
        b = min(int(g / 1000), 20)
        hist[b] = hist.get(b, 0) + 1
    with open(out, "w") as fh:
        fh.write(f"# {title}\n\nsource: `{os.path.relpath(fs[0])}`; {len(ev)} launches analysed (last 60 % of the trace)\n\n")
        fh.write(f"* span {span / 1e6:.3f} ms, inside kernels {busy / 1e6:.3f} ms = {busy / span:.3f} of the span\n")
        fh.write(f"* gaps between consecutive kernels: {len(small)} ordinary (sum {sum(small) / 1e6:.3f} ms, median {med / 1e3:.2f} us, "
                 f"mean {sum(small) / max(len(small), 1) / 1e3:.2f} us), {len(big)} long (sum {sum(big) / 1e6:.3f} ms, mean "
                 f"{sum(big) / max(len(big), 1) / 1e3:.1f} us: step boundaries = metric read-back + host start of the next step)\n")
        fh.write(f"* overlapping launches (negative gap): {sum(1 for g in gaps if g <= 0)}\n\n| gap (us) | count |\n|---|---|\n")
        for b in sorted(hist):
            fh.write(f"| {b}-{b + 1}{'+' if b == 20 else ''} | {hist[b]} |\n")
    print(open(out).read())


if __name__ == "__main__":
    main()
