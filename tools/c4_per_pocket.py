#!/usr/bin/env python
"""Per-pocket single-GPU step times of the C4 set (BASELINE config 4: 100 pockets x 100 samples) and what they predict for
the load balance of a multi-GPU run: max / mean of the per-rank work at 2 / 4 / 8 ranks under the reference's i % N round-robin
and under the opt-in size-balanced (LPT) assignment, with the measured times and with the node counts (what a rank can know
up front) as the cost.

    python tools/c4_per_pocket.py [--steps 10] > profiles/r03_c4_per_pocket.json
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from targetdiff_amd import workloads  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--steps', type=int, default=10)
    ap.add_argument('--warmup', type=int, default=2)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    model = bench.build_model(dev)
    pockets, spp, sizes, desc = bench.make_workload('c4', 0)
    ms, nodes = [], []
    for i, p in enumerate(pockets):
        pdev = workloads.DevicePocket(p, dev)
        batch = workloads.pack_samples_device(pdev, spp, sizes)
        gen = torch.Generator(device='cpu').manual_seed(2021 + i)
        lpos, lv = workloads.init_ligand(workloads.pack_samples(p, spp, sizes), generator=gen, spread=bench.LIGAND_SPREAD)
        sm = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos.to(dev),
                                  lv.to(dev), batch.ligand_element_batch, num_steps=args.steps + args.warmup, center_pos_mode='protein',
                                  max_graph_nodes=p.num_atoms + max(sizes))
        for _ in range(args.warmup):
            sm.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            sm.step()
        torch.cuda.synchronize()
        ms.append((time.perf_counter() - t0) / args.steps * 1e3)
        nodes.append(int(batch.protein_pos.shape[0] + lpos.shape[0]))
        del sm
    pred = {}
    for n in (2, 4, 8):
        pred[str(n)] = {
            'round_robin_measured_cost': workloads.predicted_imbalance(ms, n),
            'balanced_by_node_count_measured_cost': _imbalance(ms, workloads.lpt_assignment(nodes, n)),
            'balanced_by_measured_cost': workloads.predicted_imbalance(ms, n, balanced=True)}
    out = {'workload': desc, 'steps_timed': args.steps, 'ms_per_step': ms, 'nodes': nodes,
           'protein_atoms': [p.num_atoms for p in pockets], 'total_ms_per_step': sum(ms),
           'one_gpu_ligands_per_s': len(pockets) * spp / sum(ms), 'predicted_max_over_mean': pred,
           'note': 'cost of a pocket = its measured ms per step x 1000 steps; a rank\'s work = the sum over its pockets; '
                   'max/mean over ranks bounds the scaling efficiency of the strong-scaling C4 job from above (1 / value)'}
    print(json.dumps(out))


def _imbalance(cost, parts):
    sums = [sum(cost[i] for i in p) for p in parts]
    return max(sums) / (sum(sums) / len(sums))


if __name__ == '__main__':
    main()
