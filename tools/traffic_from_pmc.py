#!/usr/bin/env python3
"""profiles/traffic_x2h_{value,key}.json from a PMC summary (tools/pmc_summary.py output):
    python tools/traffic_from_pmc.py profiles/r03_pmc_c2.txt [c2|c3|c5]      (c3 / c5: traffic_x2h_*_c3.json ...)
FETCH_SIZE / WRITE_SIZE are in KiB per launch (mean over the launches of the profiled run); bench.py applies the gfx950
correction (FETCH_SIZE doubled) when it reads these files."""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def parse(path):
    kernels, cur = {}, None
    for line in open(path):
        if line.startswith('== '):
            cur = kernels.setdefault(line[3:].strip(), {})
        elif cur is not None:
            m = re.match(r'\s+(\S+)\s+mean\s+([0-9.]+)\s+\(n=(\d+)\)', line)
            if m:
                cur[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    return kernels


def main():
    src = sys.argv[1]
    workload = sys.argv[2] if len(sys.argv) > 2 else 'c2'
    suffix = '' if workload == 'c2' else '_' + workload
    kernels = parse(src)
    from targetdiff_amd import capi
    tag = capi.build_tag()          # the PMC passes and this script run in one gpurun call, on the same library file
    # x2h stage instantiations: value pass; key pass tagged STAGE = 0, not RAW
    picks = {f'traffic_x2h_value{suffix}.json': [k for k in kernels if k.startswith('edge_value16')],     # edge_value16t_kernel (12 waves) / edge_value16_kernel<..>
             f'traffic_x2h_key{suffix}.json': [k for k in kernels if re.match(r'edge_key16_kernel<false, \d+, 0', k)]}
    for fname, names in picks.items():
        if not names:
            print(f'{fname}: no matching kernel in {src}', file=sys.stderr)
            continue
        name = max(names, key=lambda k: kernels[k].get('FETCH_SIZE', (0, 0))[1])
        c = kernels[name]
        out = {'kernel': name, 'workload': f'{workload} (default bench state: ligand cloud std 2.0 A)',
               'source': f'{os.path.relpath(src, ROOT)} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, '
                         f'tools/pmc_collect.sh; mean over the {c["FETCH_SIZE"][1]} launches of the profiled run)',
               'fetch_kb': c['FETCH_SIZE'][0], 'write_kb': c['WRITE_SIZE'][0], 'build_tag': tag,
               'note': 'bytes = (2 * fetch_kb + write_kb) * 1024: FETCH_SIZE is doubled (gfx950 counts 128-B requests as 64 B, '
                       'MI355X_MICROARCH.md section HBM)'}
        with open(os.path.join(ROOT, 'profiles', fname), 'w') as f:
            json.dump(out, f)
        print(fname, name, out['fetch_kb'], out['write_kb'])


if __name__ == '__main__':
    main()
