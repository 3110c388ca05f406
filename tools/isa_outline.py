#!/usr/bin/env python3
"""Compressed schedule of one kernel in a hipcc -S listing: memory ops, waits, barriers, MFMAs, branches.
usage: isa_outline.py file.s mangled-name-substring"""
import re, sys
src, pat = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if pat in l and l.rstrip().endswith(':') or (pat in l and ': ' in l and l.startswith('_Z')))
end = next(i for i in range(start, len(lines)) if lines[i].startswith('.Lfunc_end'))
rx = re.compile(r'(v_mfma\w+|s_barrier|s_waitcnt[^;]*|ds_read\w+|ds_write\w+|buffer_load\w+|buffer_store\w+|global_load\w+|global_store\w+|scratch_\w+|s_cbranch\w+ \S+|s_branch \S+|s_setprio \S+|s_sleep \S+|^\.LBB\S+)')
prev, cnt, first = None, 0, 0
out = []
for i in range(start, end):
    m = rx.search(lines[i].strip())
    if not m: continue
    k = m.group(1).strip()
    if k == prev: cnt += 1
    else:
        if prev: out.append((first, prev, cnt))
        prev, cnt, first = k, 1, i - start
out.append((first, prev, cnt))
for f, k, c in out: print(f, k, f'x{c}' if c > 1 else '')
