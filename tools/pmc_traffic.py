#!/usr/bin/env python
"""HBM-side bytes per launch of the gather-GEMM kernels from two rocprofv3 --pmc passes over bench.py
(FETCH_SIZE in one, WRITE_SIZE in the other; MI355X_MICROARCH.md: on gfx950 FETCH_SIZE reads half of a wide coalesced
stream -> doubled; both counters are in KiB).  The launches of the LAST step in each pass are matched, in order, with the
tag sequence bench.py wrote (--kernels-out), which separates forward from data-gradient launches of the same kernel symbol.
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


if __name__ == "__main__":
    main()
