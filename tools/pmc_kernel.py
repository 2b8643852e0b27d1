#!/usr/bin/env python
"""Per-kernel means of the counters in one or more rocprofv3 --pmc output directories (counter_collection.csv).
FETCH_SIZE / WRITE_SIZE are reported in bytes (the CSV holds KiB); FETCH_SIZE is ALSO shown doubled, the gfx950 correction
for wide coalesced streaming reads (MI355X_MICROARCH.md, HBM section).
Usage: pmc_kernel.py <substring of kernel name> <dir> [<dir> ...]   (skips the first `PMC_SKIP` dispatches, default 2)"""
import csv
import glob
import json
import os
import sys


def main():
    pat, dirs = sys.argv[1], sys.argv[2:]
    skip = int(os.environ.get("PMC_SKIP", "2"))
    out = {}
    for d in dirs:
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            per = {}
            for r in csv.DictReader(open(f)):
                if pat not in r["Kernel_Name"]:
                    continue
                k = (r["Kernel_Name"].replace("(anonymous namespace)::", "").replace("void ", "").split("(")[0], r["Counter_Name"])
                per.setdefault(k, {}).setdefault(int(r["Dispatch_Id"]), 0.0)
                per[k][int(r["Dispatch_Id"])] += float(r["Counter_Value"])
            for (kn, cn), dv in per.items():
                vals = [dv[i] for i in sorted(dv)][skip:] or [dv[i] for i in sorted(dv)]
                out.setdefault(kn, {})[cn] = sum(vals) / len(vals)
                out[kn]["dispatches_" + cn] = len(vals)
    for kn, c in out.items():
        if "FETCH_SIZE" in c:
            c["fetch_bytes"] = c["FETCH_SIZE"] * 1024
            c["fetch_bytes_x2_gfx950"] = c["FETCH_SIZE"] * 2048
        if "WRITE_SIZE" in c:
            c["write_bytes"] = c["WRITE_SIZE"] * 1024
        if "TCC_HIT_sum" in c and "TCC_MISS_sum" in c:
            c["l2_hit_rate"] = c["TCC_HIT_sum"] / max(1.0, c["TCC_HIT_sum"] + c["TCC_MISS_sum"])
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
