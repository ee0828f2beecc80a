"""Time one EGNN refine-net forward (9 layers, kNN rebuilt per layer) on the C2-sized pack.  python tools/egnn_bench.py"""
import os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT)
import bench
from targetdiff_amd import workloads
from targetdiff_amd.egnn import EGNN

dev = torch.device('cuda:0')
pockets, spp, sizes, desc = bench.make_workload('c2', 0)
b = workloads.pack_samples(pockets, spp, sizes)
lpos, lv = workloads.init_ligand(b, generator=torch.Generator().manual_seed(1), spread=2.0)
# packed order of compose_context (models/common.py:120-137): per graph protein atoms, then ligand atoms
batch = torch.cat([b.protein_element_batch, b.ligand_element_batch])
idx = torch.sort(batch, stable=True).indices
mask = torch.cat([torch.zeros(b.protein_pos.shape[0], dtype=torch.bool), torch.ones(lpos.shape[0], dtype=torch.bool)])[idx]
x = torch.cat([b.protein_pos, lpos])[idx]
batch = batch[idx]
h = torch.randn(x.shape[0], 128, generator=torch.Generator().manual_seed(2))
torch.manual_seed(2021)
net = EGNN(num_layers=9, hidden_dim=128, edge_feat_dim=4, num_r_gaussian=1, k=32, cutoff_mode='knn').to(dev)
h, x, batch, mask = h.to(dev), x.to(dev), batch.to(dev), mask.to(dev)
for _ in range(3):
    net(h, x, mask, batch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(10):
    out = net(h, x, mask, batch)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / 10 * 1e3
print(f'EGNN 9 layers, N = {h.shape[0]}: {ms:.2f} ms per forward, finite = {bool(torch.isfinite(out["h"]).all())}')
