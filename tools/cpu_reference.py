#!/usr/bin/env python
"""Time the REAL reference's CPU PyTorch path (build container only: needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tools/cpu_reference.py [--threads 8] [--steps 100] [--samples 4]
        -> profiles/r03_cpu_reference_c1.json

BASELINE config 1 in full -- the 1h36 pocket (572 atoms), 4 samples with ligand sizes from the reference prior (np seed 2021),
num_steps = 100 (t = 999 .. 900) -- through the reference's own ``ScorePosNet3D.sample_diffusion``
(models/molopt_score_model.py:633-703), its unmodified model files imported by oracle/reference_loader.py (third-party ops:
oracle/shims.py), seeded weights, torch CPU fp32.  This is the "reference CPU PyTorch path timed on host cores" that
bench.py's `cpu_baseline` (the oracle restatement, kind "port", timed on the GPU box's host) stands beside:
/root/reference does not exist on the GPU box, so the reference itself can only be timed here, on this container's cores.
The restatement is timed in the same process on the same inputs, so the two can be related.
"""
import argparse
import json
import os
import platform
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--threads', type=int, default=os.cpu_count() or 1)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--samples', type=int, default=4)
    ap.add_argument('--out', default=os.path.join(ROOT, 'profiles', 'r03_cpu_reference_c1.json'))
    ap.add_argument('--no-port', action='store_true', help='skip timing oracle/restatement.py on the same inputs')
    args = ap.parse_args()
    from oracle import reference_loader, weights
    from oracle.make_golden import build_reference_model
    from targetdiff_amd import workloads
    if not reference_loader.available():
        raise SystemExit('the reference tree is not here: this tool runs in the build container only')
    ref = reference_loader.load()
    torch.set_num_threads(args.threads)
    model, sd = build_reference_model(ref)
    with np.load(os.path.join(ROOT, 'tests', 'golden', 'pocket_1h36.npz')) as z:
        pocket, sizes = workloads.Pocket(z['pos'], z['feat'].astype(np.int64), '1h36_pocket10'), z['prior_sizes_seed2021']
    b = workloads.pack_samples(pocket, args.samples, sizes[:args.samples])
    g = torch.Generator().manual_seed(0)
    lpos, lv = workloads.init_ligand(b, generator=g)
    call = lambda n: model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                                            b.ligand_element_batch, num_steps=n, center_pos_mode='protein')
    torch.manual_seed(0)
    with torch.no_grad():
        call(1)                                        # warm-up
        t0 = time.time()
        r = call(args.steps)
        wall = time.time() - t0
    assert torch.isfinite(r['pos']).all()
    n_nodes = int(b.protein_pos.shape[0] + lpos.shape[0])
    out = {'what': 'the REAL reference (models/molopt_score_model.py:633-703 via oracle/reference_loader.py), torch CPU fp32',
           'config': f'BASELINE config 1: 1h36 pocket10 x {args.samples} samples (prior sizes) x {args.steps} steps, {n_nodes} nodes',
           'host': f'{platform.processor() or platform.machine()}, {os.cpu_count()} vCPU (the build container, not the GPU box)',
           'threads': args.threads, 'torch': torch.__version__, 'wall_s': wall, 's_per_step': wall / args.steps,
           'ligands_per_s_per_1000_step_ligand': args.samples / (1000.0 * wall / args.steps)}
    if not args.no_port:
        from oracle import restatement as R
        nl = lpos.shape[0]
        noises = torch.randn(args.steps + 1, nl, 3, generator=g)
        unis = torch.rand(args.steps + 1, nl, 13, generator=g)
        pa = (sd, None, b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv, b.ligand_element_batch)
        R.sample_diffusion(*pa, num_steps=1, noises=noises, uniforms=unis)
        t0 = time.time()
        R.sample_diffusion(*pa, num_steps=args.steps, noises=noises, uniforms=unis)
        wp = time.time() - t0
        out['port_same_inputs'] = {'what': 'oracle/restatement.py (bench.py cpu_baseline kind "port")', 'wall_s': wp,
                                   's_per_step': wp / args.steps, 'reference_over_port': wall / wp}
    with open(args.out, 'w') as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == '__main__':
    main()
