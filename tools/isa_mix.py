#!/usr/bin/env python3
"""Static instruction mix of one kernel's ISA between two labels:  tools/isa_mix.py file.s kernel_symbol
Prints per basic block: label, #VALU, #MFMA, #LDS, #VMEM, #SALU, #other  (to spot what a hot loop is made of)."""
import re, sys
src, sym = sys.argv[1], sys.argv[2]
lines = open(src).read().split('\n')
start = next(i for i, l in enumerate(lines) if l.startswith(sym + ':'))
blocks, cur = [], ['entry', {}]
for l in lines[start + 1:]:
    t = l.strip()
    if t.startswith('s_endpgm'):
        break
    m = re.match(r'^(\.LBB\d+_\d+):', t)
    if m:
        blocks.append(cur); cur = [m.group(1), {}]; continue
    if not t or t.startswith(';') or t.startswith('.'):
        continue
    op = t.split()[0]
    if op.startswith('v_mfma'): k = 'mfma'
    elif op.startswith('v_'): k = 'valu'
    elif op.startswith('ds_'): k = 'lds'
    elif op.startswith(('global_', 'buffer_', 'scratch_', 'flat_')): k = 'vmem'
    elif op.startswith('s_waitcnt'): k = 'wait'
    elif op.startswith('s_'): k = 'salu'
    else: k = 'other'
    cur[1][k] = cur[1].get(k, 0) + 1
blocks.append(cur)
tot = {}
for name, d in blocks:
    n = sum(d.values())
    if n >= int(sys.argv[3]) if len(sys.argv) > 3 else 20:
        print(f'{name:12s}', ' '.join(f'{k}={v}' for k, v in sorted(d.items())))
    for k, v in d.items(): tot[k] = tot.get(k, 0) + v
print('total       ', ' '.join(f'{k}={v}' for k, v in sorted(tot.items())))
