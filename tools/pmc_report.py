#!/usr/bin/env python
"""Markdown table: per kernel symbol, launches, mean duration (rocprofv3 --stats), HBM-side bytes per launch from the
FETCH_SIZE (x2, gfx950 correction) / WRITE_SIZE passes and the L2 hit rate; below it bench.py's own per-tag table
(algorithmic flops / bytes) so the two can be set side by side.
Usage: pmc_report.py <pmc.json> <stats_dir> <bench_kernels.json>"""
import csv
import glob
import json
import os
import sys

pmc = json.load(open(sys.argv[1]))
stats = {}
for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_stats.csv"), recursive=True):
    for r in csv.DictReader(open(f)):
        stats[r["Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]] = r
print("| kernel symbol | launches | avg us | HBM-side MB / launch (fetch x2 + write) | fetch x2 MB | write MB | L2 hit |")
print("|---|---|---|---|---|---|---|")
rows = []
for k, c in pmc.items():
    st = stats.get(k, {})
    avg = float(st.get("AverageNs", 0)) / 1e3 if st else 0.0
    calls = int(st.get("Calls", 0)) if st else c.get("dispatches_FETCH_SIZE", 0)
    f2, w = c.get("fetch_bytes_x2_gfx950", 0) / 1e6, c.get("write_bytes", 0) / 1e6
    rows.append((avg * calls, f"| `{k.strip()[:90]}` | {calls} | {avg:.1f} | {f2 + w:.2f} | {f2:.2f} | {w:.2f} | {c.get('l2_hit_rate', float('nan')):.3f} |"))
for _, line in sorted(rows, reverse=True)[:40]:
    print(line)
if len(sys.argv) > 3 and os.path.exists(sys.argv[3]):
    k = json.load(open(sys.argv[3]))
    print("\nbench.py tags of the same run (HIP events; algorithmic flops):\n")
    print("| tag | launches / step | ms / step | TFLOP/s |")
    print("|---|---|---|---|")
    for r in k["kernels"][:30]:
        print(f"| {r['kernel']} | {r['launches_per_step']} | {r['ms_per_step']} | {r['tflops']} |")
