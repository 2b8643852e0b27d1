#!/usr/bin/env python
"""Basic blocks of one kernel in a hipcc -S listing with their MFMA / spill / memory instruction counts:
   tools/isa_blocks.py file.s <substring of the mangled kernel name>"""
import re
import sys
s = open(sys.argv[1]).read()
pat = sys.argv[2]
m = re.search(r'^(\S*' + re.escape(pat) + r'[^:\s]*):[^\n]*\n(.*?)\n\.Lfunc_end', s, re.S | re.M)
body = m.group(2).split('\n')
print(m.group(1), len(body), 'lines')
blocks = []
cur = ['entry', []]
for l in body:
    mm = re.match(r'^(\.LBB\d+_\d+):', l)
    if mm:
        blocks.append(cur)
        cur = [mm.group(1), []]
    else:
        cur[1].append(l)
blocks.append(cur)
for name, ls in blocks:
    c = lambda k: sum(k in x for x in ls)
    if c('v_mfma') or c('v_readlane') > 4 or c('v_writelane') > 4:
        print(f"{name:<12} {len(ls):5d} lines  mfma {c('v_mfma'):3d}  readlane {c('v_readlane'):3d}  writelane {c('v_writelane'):3d}  buffer_load {c('buffer_load'):2d}  "
              f"ds_read {c('ds_read'):2d}  ds_write {c('ds_write'):2d}  s_load {c('s_load'):2d}  barrier {c('s_barrier'):2d}  waitcnt {c('s_waitcnt'):2d}  scratch {c('scratch_'):2d}")
