#!/usr/bin/env python
"""Per-workgroup wall clock of the x2h key / value launches of one sampling step (td_debug_wg_trace): how far the slowest
workgroup of a launch is behind the mean -- the room a finer-grained row distribution could recover.

    python tools/wg_balance.py [--workload c2] > profiles/r03_wg_balance_c2.txt
"""
import argparse
import ctypes
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from targetdiff_amd import capi, workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--workload', default='c2')
    ap.add_argument('--option', action='append', default=[])
    ap.add_argument('--detail', action='store_true')
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    bargs = argparse.Namespace(knn=32, cutoff_mode='knn', radius=6.0, cap=32, fp32_node_gemms=False, option=args.option)
    model = bench.build_model(dev, bargs)
    pockets, spp, sizes, desc = bench.make_workload(args.workload, 0)
    batch = workloads.pack_samples(pockets, spp, sizes).to(dev)
    gen = torch.Generator(device='cpu').manual_seed(2021)
    lpos, lv = workloads.init_ligand(workloads.pack_samples(pockets, spp, sizes), generator=gen, spread=bench.LIGAND_SPREAD)
    sampler = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos.to(dev),
                                   lv.to(dev), batch.ligand_element_batch, num_steps=8, center_pos_mode='protein',
                                   max_graph_nodes=max(p.num_atoms for p in pockets) + max(sizes))
    for _ in range(3):
        sampler.step()
    torch.cuda.synchronize()
    slots = 9
    buf = torch.zeros(slots, 3, 256, 8, dtype=torch.int64, device=dev)
    buf[..., 2] = torch.iinfo(torch.int64).max           # atomic min of the waves' end times (unsigned compare: below 2^63)
    lib = capi.load_library()
    assert lib.td_debug_wg_trace(ctypes.c_void_p(buf.data_ptr()), slots) == 0
    sampler.step()
    torch.cuda.synchronize()
    lib.td_debug_wg_trace(None, 0)
    t = buf.cpu().numpy().astype(np.float64) / 100.0          # microseconds (100 MHz)
    print(f'# {desc}; one step, launches in order (layer 0 .. 8); times in us from the first workgroup start of the launch')
    print('# pass layer  active_wgs   mean_busy   max_busy  launch_span  max/mean  (span - mean)/span   waves: (wg end - mean wave end) / busy, '
          '(wg end - first wave end) / busy, both averaged over the workgroups')
    for p, name in enumerate(('key', 'value', 'h2x')):
        for l in range(slots):
            st, en = t[l, p, :, 0], t[l, p, :, 1]
            on = en > 0
            if not on.any():
                continue
            t0 = st[on].min()
            busy = (en - st)[on]
            span = en[on].max() - t0
            nw = 12 if p in (0, 1) else 8          # key and (round 5) value pass: 12 waves per workgroup; fused h2x: 8
            stage = (st - t[l, p, :, 4])[on]
            wmean = (en - t[l, p, :, 3] / nw)[on] / busy
            wfirst = (en - t[l, p, :, 2])[on] / busy
            print(f'  {name:5s} {l:3d} {int(on.sum()):10d} {busy.mean():11.1f} {busy.max():10.1f} {span:12.1f} {busy.max() / busy.mean():9.3f} '
                  f'{(span - busy.mean()) / span:10.3f}   {wmean.mean():8.3f} {wfirst.mean():8.3f}   staging {stage.mean():6.1f} (max {stage.max():.1f})  first entry -> last end {en[on].max() - t[l, p, :, 4][on].min():6.1f}  entries spread {np.ptp(t[l, p, :, 4][on]):5.1f}')
            if args.detail:
                b = (en - st)
                q = lambda v: ' '.join(f'{x:6.0f}' for x in np.percentile(v, [0, 10, 50, 90, 100]))
                print(f'        busy percentiles 0/10/50/90/100, workgroups 0..199: {q(b[:200])} | 200..239: {q(b[200:240])} | 240..255: {q(b[240:])}')
                print('        last 24 workgroups: ' + ' '.join(f'{x:.0f}' for x in b[232:]))
                print('        per XCD (workgroups b % 8 == x, b < 232) mean / max: ' +
                      '  '.join(f'{b[x:232:8].mean():.0f}/{b[x:232:8].max():.0f}' for x in range(8)))


    # boundaries: last workgroup end of a launch -> first workgroup entry of the next traced launch of the layer
    print('# boundaries (us): key end -> value entry | value end -> h2x entry (node projections in between) | h2x end -> next key entry')
    def first_entry(l, p):
        on = t[l, p, :, 1] > 0
        return t[l, p, :, 4][on].min() if on.any() else None
    def last_end(l, p):
        on = t[l, p, :, 1] > 0
        return t[l, p, :, 1][on].max() if on.any() else None
    for l in range(slots):
        kv = first_entry(l, 1) - last_end(l, 0) if first_entry(l, 1) is not None and last_end(l, 0) is not None else float('nan')
        vh = first_entry(l, 2) - last_end(l, 1) if first_entry(l, 2) is not None and last_end(l, 1) is not None else float('nan')
        hk = (first_entry(l + 1, 0) - last_end(l, 2)) if l + 1 < slots and first_entry(l + 1, 0) is not None and last_end(l, 2) is not None else float('nan')
        print(f'  layer {l}: {kv:7.2f} | {vh:7.2f} | {hk:7.2f}')


if __name__ == '__main__':
    main()
