#!/usr/bin/env python
"""End-to-end check of the headline metric: one complete `sample_diffusion_ligand` call (BASELINE config 2: the 1h36
pocket, 100 samples, 1000 reverse steps, one batch) through the Python driver, wall-clocked, including the on-device
trajectory record, the single device-to-host copy and the un-batching.  Also samples the session's row counts along the
trajectory (how much of the graph the ligand touches as it evolves).

    python tools/full_run.py [--samples 100] [--steps 1000] > profiles/rNN_full_run_c2.json

Weights are the seeded random initialisation of the reference architecture (no checkpoint ships with the reference), so
the ligand geometry along the trajectory is that of an untrained denoiser.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402
from targetdiff_amd import sampling, workloads  # noqa: E402
from targetdiff_amd.models import ScorePosNet3D  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=100)
    ap.add_argument('--steps', type=int, default=1000)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    pockets, spp, sizes, desc = bench.make_workload('c2', 0)
    model = ScorePosNet3D(bench.MODEL_CONFIG, workloads.PROTEIN_FEATURE_DIM, workloads.NUM_LIGAND_CLASSES)
    model.load_state_dict(bench.seeded_state_dict(model), strict=False)
    model = model.to(dev).eval()
    sizes = sizes[:args.samples]
    gen = torch.Generator(device='cpu').manual_seed(2021)

    # row statistics along a trajectory (separate, un-timed run of the stepping interface)
    batch = workloads.pack_samples(pockets, args.samples, sizes).to(dev)
    lpos, lv = workloads.init_ligand(workloads.pack_samples(pockets, args.samples, sizes), generator=gen)
    sampler = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch,
                                   lpos.to(dev), lv.to(dev), batch.ligand_element_batch, num_steps=args.steps,
                                   center_pos_mode='protein', max_graph_nodes=pockets[0].num_atoms + max(sizes))
    stats = []
    marks = sorted(set(int(v) for v in np.linspace(0, max(0, args.steps - 5), 11)))
    k = 0
    while k < args.steps:
        if k in marks and k + 5 <= args.steps:
            torch.cuda.synchronize()
            t_mark = time.perf_counter()
            for _ in range(5):
                sampler.step()
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t_mark) / 5 * 1e3
            n_all, dirty, levels = sampler.session.row_counts()
            stats.append({'step': k, 'ms_per_step': ms, 'layer0_rows': dirty, 'receptive_field_levels': levels})
            k += 5
        else:
            sampler.step()
            k += 1
    del sampler

    # the timed end-to-end calls: the first one pays one-time costs (RNG / allocator warm-up), the second is steady state
    walls = []
    finals = []
    for rep in range(3):
        torch.manual_seed(2021 + min(rep, 1))          # calls 1 and 2 share a seed: the whole run must be reproducible
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = sampling.sample_diffusion_ligand(model, pockets[0], args.samples, batch_size=args.samples, device=dev,
                                               num_steps=args.steps, ligand_num_atoms=sizes)
        torch.cuda.synchronize()
        walls.append(time.perf_counter() - t0)
        finals.append(np.concatenate([p.ravel() for p in out[0]] + [v.ravel().astype(np.float64) for v in out[1]]))
    wall = walls[-1]
    reproducible = bool(np.array_equal(finals[1], finals[2]))
    pos = out[0]
    finite = all(np.isfinite(p).all() for p in pos)
    res = {'workload': desc, 'samples': args.samples, 'steps': args.steps, 'wall_s': wall, 'wall_s_first_call': walls[0],
           'ligands_per_s_end_to_end': args.samples / wall * (1000.0 / args.steps), 'driver_time_list_s': out[6],
           'final_positions_finite': bool(finite), 'same_seed_runs_bit_identical': reproducible,
           'final_ligand_coordinate_std_A': float(np.mean([p.std(axis=0).mean() for p in pos])),
           'trajectory_row_stats': stats}
    print(json.dumps(res))


if __name__ == '__main__':
    main()
