#!/usr/bin/env python3
"""Where does a kernel spill?  Lists the basic blocks of one kernel in a hipcc -S listing that contain MFMAs, with their
scratch (spill) operations.  usage: isa_scratch.py file.s mangled-name-substring"""
import re, sys
src = open(sys.argv[1]).read().split('\n')
key = sys.argv[2]
start = next(i for i, l in enumerate(src) if l.startswith('_Z') and key in l.split(':')[0])
end = next(i for i in range(start, len(src)) if src[i].startswith('.Lfunc_end'))
blocks, cur = [], ('entry', [])
for i in range(start + 1, end):
    l = src[i].strip()
    if re.match(r'^\.LBB\S+:', l):
        blocks.append(cur)
        cur = (l.split(':')[0], [])
    elif l and not l.startswith(';'):
        cur[1].append(l)
blocks.append(cur)
tot = 0
for name, ins in blocks:
    bf = sum('v_mfma_f32_32x32x16_bf16' in x for x in ins)
    f32 = sum('v_mfma_f32_32x32x2' in x for x in ins)
    sc = [x for x in ins if x.startswith('scratch_')]
    tot += len(sc)
    if bf or f32 or sc:
        print(f"{name:14s} {len(ins):5d} instr  bf16-mfma {bf:3d}  f32-mfma {f32:3d}  scratch {len(sc):3d} "
              f"(loads {sum(x.startswith('scratch_load') for x in sc)}, stores {sum(x.startswith('scratch_store') for x in sc)})")
print('total scratch ops', tot, 'in', len(blocks), 'blocks')
