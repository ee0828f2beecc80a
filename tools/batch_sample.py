#!/usr/bin/env python
"""Sample a set of pockets across the GPUs of one node: scripts/batch_sample_diffusion.sh + scripts/sample_diffusion.py's
``__main__`` (:119-188) as one MI355X-native entry point.

    python tools/batch_sample.py --pockets DIR_OF_PDB | synthetic:100 --result_path OUT [--gpus 8]
                                 [--num_samples 100] [--num_steps 1000] [--batch_size 100] [--start_idx 0]
                                 [--checkpoint ckpt.pt] [--seed 2021]

* pocket i is sampled by rank i % world (scripts/batch_sample_diffusion.sh:15-17), from --start_idx on (:13);
* every pocket ends in ``OUT/result_{i}.pt`` with the keys scripts/sample_diffusion.py:175-182 saves (consumed by
  scripts/evaluate_diffusion.py:70-76), written by a background thread; existing files are skipped, so a re-run
  continues where an interrupted one stopped;
* ``--gpus N`` with no launcher around the script starts N ranks itself (one process per GPU); under
  ``python -m torch.distributed.run`` it uses the ranks it is given.  No data-path collective: RCCL carries one barrier
  and one gather of per-rank timings.

``--checkpoint``: a reference checkpoint (``{'config': ..., 'model': state_dict}``, scripts/train_diffusion.py:222-228;
loaded with strict=True as scripts/sample_diffusion.py:163 does).  Without it the seeded random initialisation of the
reference architecture is used (no checkpoint ships with the reference).  Ligand sizes: the reference prior when
``utils.evaluation.atom_num`` is importable (i.e. inside the reference repo), else ``--ligand_atoms``.
"""
import argparse
import glob
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from targetdiff_amd import launch, sampling, workloads  # noqa: E402


def load_pockets(spec: str):
    if spec.startswith('synthetic:'):
        n = int(spec.split(':', 1)[1])
        return workloads.synthetic_test_set(n)
    files = sorted(glob.glob(os.path.join(spec, '*.pdb')))
    if not files:
        raise SystemExit(f'no *.pdb under {spec}')
    return [workloads.pocket_from_pdb(f, os.path.basename(f)) for f in files]


def build_model(args, dev):
    import bench
    from targetdiff_amd.models import ScorePosNet3D
    if args.checkpoint:
        ckpt = torch.load(args.checkpoint, map_location='cpu', weights_only=False)
        cfg = ckpt['config']
        cfg = cfg['model'] if isinstance(cfg, dict) else cfg.model
        model = ScorePosNet3D(cfg, workloads.PROTEIN_FEATURE_DIM, workloads.NUM_LIGAND_CLASSES)
        model.load_state_dict(ckpt['model'])                          # strict, scripts/sample_diffusion.py:163
    else:
        model = ScorePosNet3D(bench.MODEL_CONFIG, workloads.PROTEIN_FEATURE_DIM, workloads.NUM_LIGAND_CLASSES)
        model.load_state_dict(bench.seeded_state_dict(model), strict=False)
    return model.to(dev).eval()


def main(argv=None, model_factory=build_model):
    ap = argparse.ArgumentParser()
    ap.add_argument('--pockets', required=True)
    ap.add_argument('--result_path', required=True)
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--num_samples', type=int, default=100)          # configs/sampling.yml:7
    ap.add_argument('--num_steps', type=int, default=1000)           # configs/sampling.yml:8
    ap.add_argument('--batch_size', type=int, default=100)           # scripts/sample_diffusion.py:124
    ap.add_argument('--start_idx', type=int, default=0)
    ap.add_argument('--checkpoint', default=None)
    ap.add_argument('--seed', type=int, default=2021)                # configs/sampling.yml:6
    ap.add_argument('--ligand_atoms', type=int, default=0, help='fixed ligand size (0: the reference prior)')
    ap.add_argument('--balance', action='store_true', help='size-balanced pocket -> rank assignment (LPT on protein atom count) '
                    'instead of the reference\'s i %% N round-robin')
    ap.add_argument('--pin-devices', action='store_true', help='self-spawned ranks see one GPU each (HIP_VISIBLE_DEVICES=r), the '
                    'reference\'s CUDA_VISIBLE_DEVICES recipe, instead of all GPUs + set_device(LOCAL_RANK)')
    ap.add_argument('--overlap-batches', action='store_true', help='advance the sample batches of a pocket together, one HIP stream and one '
                    'captured hipGraph each (pays when batch_size is small; per-batch generators, see sample_diffusion_ligand)')
    ap.add_argument('--device', default='cuda', help="'cuda' (rank r uses GPU LOCAL_RANK) or 'cpu' (tests: gloo + stub model)")
    args = ap.parse_args(argv)
    if argv is None:
        if args.pin_devices:
            os.environ['TD_PIN_DEVICES'] = '1'
        launch.self_spawn_if_needed(args.gpus)

    rank, world = int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    on_gpu = args.device == 'cuda'
    if on_gpu:
        torch.cuda.set_device(local_rank)
        dev = torch.device('cuda', local_rank)
    else:
        dev = torch.device('cpu')
    import torch.distributed as dist
    if world > 1:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if on_gpu:
            with launch.stdout_to_stderr():          # RCCL's version banner goes to stdout
                dist.init_process_group('nccl', device_id=dev)          # RCCL; rendezvous, barrier and one gather only
                dist.barrier()
        else:
            dist.init_process_group('gloo')
    torch.manual_seed(args.seed + rank)                              # utils/misc.py:58 seed_all, decorrelated per rank
    np.random.seed(args.seed + rank)

    pockets = load_pockets(args.pockets)
    model = model_factory(args, dev)
    sizes = [args.ligand_atoms] * args.num_samples if args.ligand_atoms > 0 else None
    log = []

    def on_pocket(idx, seconds, skipped):
        log.append({'pocket': idx, 'seconds': seconds, 'skipped': skipped})
        print(f'[rank {rank}] pocket {idx}: ' + ('exists, skipped' if skipped else f'{seconds:.2f} s'), file=sys.stderr, flush=True)
    t0 = time.time()
    sampling.run_sharded(model, pockets, args.num_samples, rank=rank, world_size=world, start_idx=args.start_idx,
                         result_path=args.result_path, keep_results=False, on_pocket=on_pocket, balance=args.balance,
                         batch_size=args.batch_size, device=dev, num_steps=args.num_steps, ligand_num_atoms=sizes,
                         **({'overlap_batches': True} if args.overlap_batches else {}))
    if on_gpu:
        torch.cuda.synchronize()
    meta = {'rank': rank, 'wall_s': time.time() - t0, 'pockets': log}
    gathered = sampling.gather_metadata(meta)
    if world > 1:
        dist.barrier()
    if rank == 0:
        walls = [m['wall_s'] for m in gathered]
        done = sum(1 for m in gathered for p in m['pockets'] if not p['skipped'])
        summary = {'world_size': world, 'pockets_sampled': done,
                   'pockets_skipped': sum(1 for m in gathered for p in m['pockets'] if p['skipped']),
                   'ligands': done * args.num_samples, 'wall_s': max(walls),
                   'ligands_per_s': done * args.num_samples / max(max(walls), 1e-9),
                   'load_imbalance_max_over_mean': max(walls) / max(sum(walls) / len(walls), 1e-9), 'per_rank': gathered}
        os.makedirs(args.result_path, exist_ok=True)
        with open(os.path.join(args.result_path, 'summary.json'), 'w') as f:
            json.dump(summary, f, indent=1)
        print(json.dumps({k: v for k, v in summary.items() if k != 'per_rank'}))
    if world > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
