#!/usr/bin/env python
"""Per-node phase timeline (cycle stamps) of workgroup 0 of one x2h key-pass launch on the C2 workload."""
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
NODES = 32
buf = torch.zeros(8 * NODES * 8, dtype=torch.int64, device=dev)
lib = capi.load_library()
lib.td_debug_edge_timing(c_void_p(buf.data_ptr()), NODES)
sampler.step()
torch.cuda.synchronize()
lib.td_debug_edge_timing(None, 0)
a = buf.cpu().numpy().reshape(8, NODES, 8)
t0 = a[:, :, 0][a[:, :, 0] > 0].min()
names = ['geo+gather issue', 'exp+MFMA1', 'LayerNorm', 'transpose', 'MFMA2', 'epilogue']
for w in (0, 4, 1):
    print(f'--- wave {w}: node start (rel. to kernel start) and phase lengths [cycles]: {names}')
    for nd in range(8, 14):
        st = a[w, nd]
        print(f'   node {nd}: start {int(st[0] - t0):8d}  phases {[int(st[k + 1] - st[k]) for k in range(6)]}  total {int(st[6] - st[0])}')
ph = np.diff(a[:, 4:28, :7], axis=2).reshape(-1, 6)
print('mean phase lengths over waves/nodes:', dict(zip(names, ph.mean(0).round().astype(int).tolist())), 'total', int(ph.sum(1).mean()))
