#!/usr/bin/env python
"""Dump per-segment cycle stamps of workgroup 0 of one x2h launch (C2 workload)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from targetdiff_amd import capi, workloads
from targetdiff_amd.models import ScorePosNet3D
from ctypes import c_void_p

dev = torch.device('cuda:0')
pockets, spp, sizes, desc = bench.make_workload('c2', 0)
model = ScorePosNet3D(bench.MODEL_CONFIG, 27, 13)
model.load_state_dict(bench.seeded_state_dict(model), strict=False)
model = model.to(dev).eval()
batch = workloads.pack_samples(pockets, spp, sizes).to(dev)
lpos, lv = workloads.init_ligand(workloads.pack_samples(pockets, spp, sizes), generator=torch.Generator().manual_seed(1))
sampler = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch,
                               lpos.to(dev), lv.to(dev), batch.ligand_element_batch, num_steps=3, center_pos_mode='protein')
sampler.step()
SEGS = 64
buf = torch.zeros(8 * SEGS * 8, dtype=torch.int64, device=dev)
lib = capi.load_library()
lib.td_debug_edge_timing(c_void_p(buf.data_ptr()), SEGS)
sampler.step()
torch.cuda.synchronize()
lib.td_debug_edge_timing(None, 0)
a = buf.cpu().numpy().reshape(8, SEGS, 8)
t0 = a[:, :, 0][a[:, :, 0] > 0].min()
np.set_printoptions(linewidth=200)
for seg in range(20, 28):
    print(f'--- segment {seg}')
    for w in (0, 4):
        st = a[w, seg]
        rel = [int(x - a[w, seg, 0]) if x > 0 else -1 for x in st[:6]]
        print(f'  wave {w} ({"k" if w < 4 else "v"}-role): start@{int(st[0] - t0):8d}  stamps(rel) {rel}')
seg_len = np.diff(a[0, 10:60, 0])
print('segment length (wave 0) cycles: mean', seg_len.mean(), 'even', seg_len[0::2].mean(), 'odd', seg_len[1::2].mean())
