#!/usr/bin/env python3
"""Instruction mix per basic block of one kernel in a `hipcc -S --cuda-device-only` listing.
usage: isa_blocks.py <listing.s> <kernel name prefix> [out.s]"""
import re, sys
s = open(sys.argv[1]).read().split('\n')
start = next(i for i, l in enumerate(s) if l.startswith(sys.argv[2]) and ':' in l)
end = next(i for i in range(start, len(s)) if s[i].strip().startswith('.Lfunc_end'))
body = s[start:end]
if len(sys.argv) > 3:
    open(sys.argv[3], 'w').write('\n'.join(body))
def cls(op):
    if op.startswith('v_'): return 'V'
    if op.startswith('s_'): return 'S'
    if op.startswith('ds_'): return 'L'
    if op.startswith(('global_', 'buffer_', 'flat_', 'scratch_')): return 'M'
    return '?'
blocks, cur = [], ['entry', {}, []]
for l in body[1:]:
    t = l.strip()
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), {}, []]
    elif t and not t.startswith((';', '.')):
        op = t.split()[0]
        cur[1][cls(op)] = cur[1].get(cls(op), 0) + 1
        if op.startswith(('s_cbranch', 's_branch')): cur[2].append(t.split()[0][2:] + '->' + t.split()[-1])
blocks.append(cur)
tot = {}
for name, c, br in blocks:
    for k, v in c.items(): tot[k] = tot.get(k, 0) + v
    print("%-12s V%4d S%4d L%3d M%3d  %s" % (name, c.get('V', 0), c.get('S', 0), c.get('L', 0), c.get('M', 0), ' '.join(br)))
print("total", tot)
