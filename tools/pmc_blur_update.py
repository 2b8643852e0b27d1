#!/usr/bin/env python
"""Rewrites the blur256 entry of profiles/hbm_traffic.json from tools/pmc_blur.sh outputs (gpurun_out/pmc_<tag>.json), tagging it
with the hash of the blur source they were taken on.  Usage: pmc_blur_update.py <taps>=<pmc json> [...]   e.g. 31=gpurun_out/pmc_f31.json"""
import hashlib
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "hbm_traffic.json")
d = json.load(open(path))
srcs = ["blurred-gan_amd/csrc/blur.hip", "blurred-gan_amd/csrc/blur_panel.hip", "blurred-gan_amd/csrc/blur_panel.h"]
h = hashlib.sha1()
for s_ in srcs:
    h.update(open(os.path.join(root, s_), "rb").read())
sha = h.hexdigest()[:16]
kernels = {}
for arg in sys.argv[1:]:
    taps, f = arg.split("=")
    pm = json.load(open(f))
    names = [k for k in pm if "blur" in k]
    fetch = sum(pm[k]["fetch_bytes_x2_gfx950"] for k in names) / len(names)
    write = sum(pm[k]["write_bytes"] for k in names) / len(names)
    e = {"launches_per_application": 2 if len(names) == 1 and "band" in names[0] else len(names), "kernel": ", ".join(names),
         "fetch_bytes_per_launch": fetch, "write_bytes_per_launch": write, "hbm_bytes_per_launch": fetch + write}
    hit = [pm[k]["l2_hit_rate"] for k in names if "l2_hit_rate" in pm[k]]
    if hit:
        e["l2_hit_rate"] = sum(hit) / len(hit)
    kernels[f"blur{taps}"] = e
d["entries"] = [e for e in d["entries"] if e.get("arch") != "blur256"]
d["entries"].append({"arch": "blur256", "batch": 64, "sources": srcs, "sources_sha": sha, "kernels": kernels})
json.dump(d, open(path, "w"), indent=1)
print(json.dumps(kernels, indent=1))
