#!/usr/bin/env python
"""HBM-side bytes per launch of the gather-GEMM kernels from two rocprofv3 --pmc passes over bench.py
(FETCH_SIZE in one, WRITE_SIZE in the other; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half of a wide coalesced
stream -> doubled; both counters are in KiB).  The launches of the LAST step in each pass are matched, in order, with the
tag sequence bench.py wrote (--kernels-out), which separates forward from data-gradient launches of the same kernel symbol.
Also writes <out.json>.wgrad.md: every filter-gradient launch of the step (matched the same way) with its ALGORITHMIC bytes (x and
dy read once, dw written once: the figure the library records) beside the counted ones and their ratio.
Usage: pmc_traffic.py <fetch_dir> <write_dir> <kernels.json> <out.json>"""
import csv
import glob
import json
import os
import sys


def dispatches(d, counter):
    f = glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True)
    assert f, f"no counter_collection.csv under {d}"
    assert len(f) == 1, f"{len(f)} counter files under {d}: profile into a fresh directory"
    rows = {}
    for r in csv.DictReader(open(f[0])):
        if r["Counter_Name"] != counter:
            continue
        key = int(r["Dispatch_Id"])
        rows[key] = (r["Kernel_Name"], rows.get(key, ("", 0.0))[1] + float(r["Counter_Value"]))
    return [rows[k] for k in sorted(rows)]


def main():
    fdir, wdir, kjson, out = sys.argv[1:5]
    seq = [s["kernel"] for s in json.load(open(kjson))["sequence"]]
    tags = [t for t in seq if t in ("conv_igemm_fwd", "conv_igemm_dgrad")]
    res = {}
    for name, d, counter, mul in (("fetch", fdir, "FETCH_SIZE", 2.0), ("write", wdir, "WRITE_SIZE", 1.0)):
        ds = [v for n, v in dispatches(d, counter) if "conv_igemm_kernel" in n]
        assert len(ds) >= len(tags), (len(ds), len(tags))
        last = ds[-len(tags):]
        for t, v in zip(tags, last):
            res.setdefault(t, {}).setdefault(name, []).append(v * 1024.0 * mul)
    outd = {}
    for t, d in res.items():
        n = len(d["fetch"])
        outd[t] = {"launches_per_step": n, "fetch_bytes_per_launch": sum(d["fetch"]) / n, "write_bytes_per_launch": sum(d["write"]) / n,
                   "hbm_bytes_per_launch": (sum(d["fetch"]) + sum(d["write"])) / n}
    json.dump(outd, open(out, "w"), indent=1)
    print(json.dumps(outd, indent=1))
    # ---- filter gradients, launch by launch
    full = json.load(open(kjson))["sequence"]
    wtags = [e for e in full if e["kernel"].startswith("conv_wgrad") and "reduce" not in e["kernel"]]
    isw = lambda n: "conv_wgrad" in n and "reduce" not in n
    fd = [(n, v) for n, v in dispatches(fdir, "FETCH_SIZE") if isw(n)]
    wd = [(n, v) for n, v in dispatches(wdir, "WRITE_SIZE") if isw(n)]
    if wtags and len(fd) >= len(wtags) and len(wd) >= len(wtags):
        fd, wd = fd[-len(wtags):], wd[-len(wtags):]
        with open(out + ".wgrad.md", "w") as fh:
            fh.write("| launch (step order) | kernel symbol | us | algorithmic MB | HBM-side MB (fetch x2 + write) | ratio |\n|---|---|---|---|---|---|\n")
            for i, (e, (fn, fv), (_, wv)) in enumerate(zip(wtags, fd, wd)):
                hbm = (fv * 2.0 + wv) * 1024.0 / 1e6
                sym = fn.replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0]
                alg = e.get("alg_mb") or 0.0
                fh.write(f"| {i} {e['kernel']} ({e['gflop']} GFLOP) | `{sym[:60]}` | {e['us']} | {alg:.1f} | {hbm:.1f} | {hbm / alg if alg else float('nan'):.2f} |\n")
        print(open(out + ".wgrad.md").read())
    # ---- gather-GEMM launches, launch by launch (round 4): where the dominant kernel's excess over its algorithmic bytes sits
    gtags = [e for e in full if e["kernel"] in ("conv_igemm_fwd", "conv_igemm_dgrad")]
    isg = lambda n: "conv_igemm_kernel" in n
    fd = [(n, v) for n, v in dispatches(fdir, "FETCH_SIZE") if isg(n)]
    wd = [(n, v) for n, v in dispatches(wdir, "WRITE_SIZE") if isg(n)]
    if gtags and len(fd) >= len(gtags) and len(wd) >= len(gtags):
        fd, wd = fd[-len(gtags):], wd[-len(gtags):]
        with open(out + ".igemm.md", "w") as fh:
            fh.write("| launch (step order) | tile | us | GFLOP | algorithmic MB | fetch x2 MB | write MB | HBM-side MB | ratio |\n|---|---|---|---|---|---|---|---|---|\n")
            for i, (e, (fn, fv), (_, wv)) in enumerate(zip(gtags, fd, wd)):
                f2, w = fv * 2.0 * 1024.0 / 1e6, wv * 1024.0 / 1e6
                tile = fn.split("conv_igemm_kernel<")[1].split(">")[0] if "conv_igemm_kernel<" in fn else "?"
                alg = e.get("alg_mb") or 0.0
                fh.write(f"| {i} {e['kernel']} | {tile} | {e['us']} | {e['gflop']} | {alg:.1f} | {f2:.1f} | {w:.1f} | {f2 + w:.1f} | {(f2 + w) / alg if alg else float('nan'):.2f} |\n")


if __name__ == "__main__":
    main()
