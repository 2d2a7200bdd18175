#!/usr/bin/env python3
"""Weighted VALU cost of the basic blocks of a kernel in a gfx950 assembly listing (measurement tooling).

    hipcc --offload-arch=gfx950 -O3 -S --cuda-device-only file.hip -o file.s; python scripts/isa_cost.py file.s <kernel-name-substring>
"""
import re, collections, sys
COST = [('v_exp',7.1),('v_rcp',7.1),('v_log',7.1),('v_sqrt',7.1),('v_rsq',7.1),('v_permlane',7.1),('v_pk_fma',4.3),('v_pk_',3.8),('v_cmp',3.8),('v_min',3.7),('v_max',3.7),
        ('v_cndmask',3.8),('v_fma',2.6),('v_fmac',2.6),('v_mov_b64',4.4),('v_readlane',3.8),('v_readfirstlane',3.8)]
def cost(ins, line):
    if not ins.startswith('v_'): return 0.0
    if 'dpp' in ins or 'row_' in line or 'quad_perm' in line: return 3.8
    for k, c in COST:
        if ins.startswith(k): return c
    return 2.2
s = open(sys.argv[1]).read()
pat = sys.argv[2]
names = [n for n in re.findall(r'^(_ZN3bds\S*):', s, re.M) if pat in n]
for nm in names:
    body = s[s.index(nm+':'):]
    body = body[:body.index('.Lfunc_end')]
    cur=None; blocks=collections.OrderedDict()
    for l in body.split('\n'):
        l=l.strip()
        if re.match(r'^\.LBB\d+_\d+:', l):
            cur=l.split(':')[0]; blocks[cur]=[]
        elif cur and l and not l.startswith(';') and not l.startswith('.'):
            blocks[cur].append(l)
    print(nm[:70])
    for b,ls in blocks.items():
        ins=[l.split()[0] for l in ls]
        v=sum(1 for i in ins if i.startswith('v_'))
        cyc=sum(cost(l.split()[0], l) for l in ls)
        if v>=12:
            c=collections.Counter(ins)
            print('  %-10s n=%3d valu=%3d cyc=%6.1f' % (b,len(ins),v,cyc), dict(c.most_common(9)))
    for m in re.finditer(r'; (NumVgprs|Occupancy|ScratchSize): (\d+)', body): print('  ', m.group(0))
