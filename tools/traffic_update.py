#!/usr/bin/env python
"""Rewrites one training-step entry of profiles/hbm_traffic.json from tools/pmc_step.sh's pmc_<tag>_traffic.json, tagging it
with the hash of the gather-GEMM sources it was measured on (bench.py reports an entry only while they still match).
Usage: traffic_update.py <arch> <batch> <pmc_<tag>_traffic.json>"""
import hashlib
import json
import os
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path = os.path.join(root, "profiles", "hbm_traffic.json")
arch, batch, src = sys.argv[1], int(sys.argv[2]), sys.argv[3]
sources = ["blurred-gan_amd/csrc/conv_igemm.hip", "blurred-gan_amd/csrc/conv_common.h"]
h = hashlib.sha1()
for s in sources:
    h.update(open(os.path.join(root, s), "rb").read())
d = json.load(open(path))
d["entries"] = [e for e in d["entries"] if not (e.get("arch") == arch and e.get("batch") == batch)]
d["entries"].insert(0, {"arch": arch, "batch": batch, "sources": sources, "sources_sha": h.hexdigest()[:16], "kernels": json.load(open(src))})
json.dump(d, open(path, "w"), indent=1)
print("updated", arch, batch, h.hexdigest()[:16])
