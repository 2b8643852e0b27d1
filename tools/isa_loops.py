#!/usr/bin/env python
"""Self-loops of every kernel in hipcc -S listings whose body holds at most three global / buffer loads and a `s_waitcnt vmcnt(0)`:
one round trip per iteration unless other waves cover it (profiles/r05_a_gather_gemm_limits.md section 12).
   for f in blurred-gan_amd/csrc/*.hip; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only $f -o /tmp/$(basename $f .hip).s; done
   tools/isa_loops.py /tmp/*.s"""
import re
import sys
for f in sys.argv[1:]:
    s = open(f).read()
    for m in re.finditer(r'^(_Z\S*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        parts = re.split(r'\n(\.LBB\d+_\d+):[^\n]*', body)
        for i in range(1, len(parts), 2):
            lab, txt = parts[i], parts[i + 1]
            if not re.search(r's_cbranch_\w+\s+' + re.escape(lab) + r'\b', txt):
                continue
            lines = [l.strip() for l in txt.split('\n')]
            loads = sum(('global_load' in l or 'buffer_load' in l) for l in lines)
            w0 = sum('vmcnt(0)' in l for l in lines)
            if 1 <= loads <= 3 and w0 >= 1:
                short = re.sub(r'^_ZN\d+_GLOBAL__N_1\d+', '', name)[:80]
                print(f"{f.split('/')[-1]:18s} {short:82s} {lab:10s} loads {loads} vmcnt(0) {w0} mfma {sum('v_mfma' in l for l in lines)} lines {len(lines)}")
