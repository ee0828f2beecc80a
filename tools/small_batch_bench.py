#!/usr/bin/env python
"""The driver in the small-batch regime: `sample_diffusion_ligand(model, data, 96, batch_size=16)` (the reference
signature's default batch size, scripts/sample_diffusion.py:31) on the 1h36 pocket, sample batches one after the other
vs advanced together on one HIP stream each (`overlap_batches=True`).

    python tools/small_batch_bench.py [--samples 96] [--batch-size 16] [--steps 200] > profiles/rNN_small_batch.json
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
from targetdiff_amd import sampling  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--samples', type=int, default=96)
    ap.add_argument('--batch-size', type=int, default=16)
    ap.add_argument('--steps', type=int, default=200)
    args = ap.parse_args()
    dev = torch.device('cuda:0')
    torch.cuda.set_device(dev)
    model = bench.build_model(dev)
    pocket, sizes = bench.load_1h36()
    sizes = [int(v) for v in sizes[:args.samples]]
    out = {'workload': f'1h36 pocket10, {args.samples} samples in batches of {args.batch_size}, {args.steps} steps'}
    # one untimed batch first: code objects load and dynamic-LDS attributes are set on a kernel's first launch
    sampling.sample_diffusion_ligand(model, pocket, args.batch_size, batch_size=args.batch_size, device=dev, num_steps=3,
                                     ligand_num_atoms=sizes[:args.batch_size])
    for name, ov, ug in (('sequential', False, None), ('overlapped', True, None), ('overlapped_launch_by_launch', True, False),
                         ('sequential_again', False, None)):
        torch.manual_seed(2021)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        res = sampling.sample_diffusion_ligand(model, pocket, args.samples, batch_size=args.batch_size, device=dev,
                                               num_steps=args.steps, ligand_num_atoms=sizes, overlap_batches=ov, use_graph=ug)
        torch.cuda.synchronize()
        wall = time.perf_counter() - t0
        nb = len(res[6])
        out[name] = {'wall_s': wall, 'batches': nb, 'ms_per_batch_step': wall / (nb * args.steps) * 1e3,
                     'ligands_per_s_at_1000_steps': args.samples / wall * args.steps / 1000.0}
    out['speedup'] = out['sequential_again']['wall_s'] / out['overlapped']['wall_s']
    out['graph_replay_gain_in_overlapped_mode'] = out['overlapped_launch_by_launch']['wall_s'] / out['overlapped']['wall_s']
    print(json.dumps(out))


if __name__ == '__main__':
    main()
