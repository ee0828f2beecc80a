#!/usr/bin/env python
"""bench.py -- ligands/sec of the TargetDiff denoising hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload c2|c3|c1|c5|c4] [--no-cpu-baseline] [--no-full-run]

`--gpus N` with no launcher around the script (no RANK / WORLD_SIZE in the environment) starts the N ranks itself, one
process per GPU (targetdiff_amd/launch.py; the environment each rank sees is the one `python -m torch.distributed.run
--nnodes=1 --nproc-per-node N` provides, so both launchers run the same code path).

A "step" is one reverse-diffusion step (denoiser forward + posterior update + on-device trajectory
record) over one packed batch.  Default workload = BASELINE.json configs[1] (the configuration the
metric is quoted on): the 1h36 pocket (572 protein atoms), 100 samples with ligand sizes from the
reference prior (np seed 2021), packed in one ragged graph on one GPU.  Inputs are resident in HBM
before the timed region.  The timed steps start from a ligand cloud of std 2.0 A per coordinate -- the geometry a
1000-step run spends its time in (see LIGAND_SPREAD below; the step time depends on it because the sampling session
skips rows the ligand cannot influence); `--initial-state` additionally times the sampler's initial state N(0, I)
(`initial_state`; off by default so that a profiler attached to the default command sees one workload only).  metric value = n_gpus * samples_per_batch / (1000 steps * seconds_per_step):
the rate at which finished ligands leave a 1000-step sampler.  With N > 1 (torch.distributed.run, one
rank per GPU over RCCL) every rank samples its own pocket replica -- pockets shard with no data-path
collective (scripts/batch_sample_diffusion.sh:15-20) -- so scaling is "weak".

Extra objects on the JSON line:
  roofline     dominant kernel = edge_value16t_kernel (x2h value pass; the key pass edge_key16_kernel is its twin and is
               reported under roofline.key_pass);
               achieved = executed algorithmic FLOPs per launch (327,680 per dst node) / mean
               launch time from HIP events recorded on the launch stream inside the timed region;
               peak = 157.3 TFLOP/s, the dense fp32 MFMA peak (= fp32 vector peak) -- the dtype the path computes in.
               matrix_bound_as_built is reported beside it: with the radial/type first layer on bf16 piece triples
               (default) that layer's 4 x 32-deep K-packed bf16 products (the six piece products of the 21 inputs fill four
               instructions' K slots since round 5) are priced at the 2.5 PFLOP/s dense bf16 peak and everything else at
               the fp32 peak (= 224 TFLOP/s of the same algorithmic FLOPs for a 32-edge row).
  cpu_baseline the oracle restatement (torch CPU, same weights) timed on this host's cores over a bounded
               sample of the same workload (the same pocket, fewer samples, a few steps).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from targetdiff_amd import capi, launch, workloads  # noqa: E402
from targetdiff_amd.models import ScorePosNet3D  # noqa: E402

PEAK_FP32_MFMA_TFLOPS = 157.3          # /opt/skills/guides/MI355X_MICROARCH.md (dense fp32 MFMA = vector peak)
PEAK_BF16_MFMA_TFLOPS = 2500.0         # same guide: ~2.5 PFLOP/s dense bf16 (2495 measured)
# first layer of an attention pass on bf16 piece triples: the six piece products of the 21 inputs K-packed into four 32-deep instructions
# (csrc/edge16.hip td_pk4_tiles), all 32 slots of the row
FIRST_LAYER_FLOP_BF16_EXECUTED = 4 * 2 * 32 * 128 * 32
FIRST_LAYER_FLOP_F16_EXECUTED = 2 * 2 * 32 * 128 * 32     # f16 piece pairs: two 32-deep instructions per tile
PEAK_HBM_GBS = 8000.0                  # same guide: HBM3E ~ 8 TB/s
# Dominant kernels: edge_value16t_kernel (x2h value pass, default graph) and its twin edge_key16_kernel (x2h key pass).  FLOPs per dst node,
# identical for the two passes:
#   executed  = 2 * (32*128*20 [radial/type first layer] + 128*128 [out = W2v Zbar | U_i = W2k^T q_i]
#                    + 32*128*16 [Zbar = alpha^T z | logits = z U_i]) = 327,680
#   canonical = 2 * (32*128*20 + 32*128*128 [per-edge second Linear] + 32*128 [alpha.v | q.k]) = 1,220,608  (SURVEY.md
#               section 8d per-edge figures x 32 edges: what the reference formulation spends on the same stage)
# `roofline.achieved` uses the executed count (so frac <= 1 is meaningful for this kernel); the canonical figure is
# reported next to it.  DESIGN.md section 4 derives both.
KEY_PASS_FLOP_EXECUTED = 2 * (32 * 128 * 20 + 128 * 128 + 32 * 128 * 16)
KEY_PASS_FLOP_CANONICAL = 2 * (32 * 128 * 20 + 32 * 128 * 128 + 32 * 128)
# Geometry the timed steps run on.  The session skips rows the ligand cannot influence, so the step time depends on how far
# the ligand cloud reaches into the pocket.  A 1000-step run starts from N(0, I) around the pocket centre (std 1 A) and,
# following the forward marginals std_t^2 = abar_t * std_0^2 + (1 - abar_t) with std_0 ~= Rg / sqrt(3) ~= 2.0 .. 2.3 A for
# a 20-30 heavy-atom ligand and abar_999 ~= 0.37 (sigmoid schedule), spends nearly all of its steps at std 1.6 .. 2.2 A.
# The headline number is therefore timed at std 2.0 A; --initial-state times the (cheaper) initial state next to it.
LIGAND_SPREAD = 2.0
METRIC = 'ligands/sec (1000-step sampling, 100 samples/pocket) at 1/2/4/8 MI355X'

# configs/training.yml:9-42
MODEL_CONFIG = dict(
    model_mean_type='C0', beta_schedule='sigmoid', beta_start=1.e-7, beta_end=2.e-3, v_beta_schedule='cosine',
    v_beta_s=0.01, num_diffusion_timesteps=1000, loss_v_weight=100., sample_time_method='symmetric', time_emb_dim=0,
    time_emb_mode='simple', center_pos_mode='protein', node_indicator=True, model_type='uni_o2', num_blocks=1,
    num_layers=9, hidden_dim=128, n_heads=16, edge_feat_dim=4, num_r_gaussian=20, knn=32, num_node_types=8,
    act_fn='relu', norm=True, cutoff_mode='knn', ew_net_type='global', num_x2h=1, num_h2x=1, r_max=10.,
    x2h_out_fc=False, sync_twoup=False)


def load_1h36():
    with np.load(os.path.join(ROOT, 'tests', 'golden', 'pocket_1h36.npz')) as z:
        return workloads.Pocket(z['pos'], z['feat'].astype(np.int64), '1h36_pocket10'), z['prior_sizes_seed2021']


def make_workload(name: str, rank: int):
    """-> (pockets, samples_per_pocket, ligand sizes per graph, description)"""
    if name == 'c2':
        pocket, sizes = load_1h36()
        return [pocket], 100, [int(s) for s in sizes], 'C2: 1h36 pocket10 (572 atoms) x 100 samples, prior sizes'
    if name == 'c1':
        pocket, sizes = load_1h36()
        return [pocket], 4, [int(s) for s in sizes[:4]], 'C1: 1h36 pocket10 x 4 samples'
    if name == 'c3':
        pockets = [workloads.synthetic_pocket(1000 + p + 32 * rank, 300) for p in range(32)]
        return pockets, 100, [25] * 3200, 'C3: 32 synthetic 300-atom pockets x 100 samples x 25 ligand atoms'
    if name == 'c5':
        pocket = workloads.synthetic_pocket(5000 + rank, 1000, 4.0, 21.0)
        return [pocket], 256, [30] * 256, 'C5: synthetic 1000-atom pocket x 256 samples x 30 ligand atoms'
    if name == 'c4':
        # BASELINE config 4: 100 pockets x 100 samples, pocket i -> rank i % world (scripts/batch_sample_diffusion.sh:15-20);
        # synthetic stand-ins sized 250-600 atoms (the CrossDocked test split is not available), ligand size 25
        pockets = workloads.synthetic_test_set(100)
        return pockets, 100, [25] * 100, 'C4: 100 synthetic pockets (250-600 atoms) x 100 samples x 25 ligand atoms, pocket i -> rank i % N'
    raise ValueError(name)


def geometry_sweep(model, pocket, sizes, dev, steps=10):
    """The session's step time depends on how far the ligand reaches into the pocket (rows no ligand atom can influence are not
    recomputed): time `steps` session steps from Gaussian clouds of 1 / 2 / 3 / 4 A per coordinate around the pocket centroid (1 A =
    the sampler's own initial state, 2 A = the headline's state) and from the DOCKED pose of the reference's example ligand
    (examples/1h36_A_rec_1h36_r88_lig_tt_docked_0.sdf, 25 atoms, replicated per sample with 0.25 A of noise: the geometry a trained
    model's trajectory ends in).  The stateless floor is geometry-independent (reported beside it)."""
    out = {}
    n = len(sizes)

    def run(batch, lpos, lv, max_nodes):
        b = batch.to(dev)
        s = model.begin_sampling(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos.to(dev), lv.to(dev),
                                 b.ligand_element_batch, num_steps=steps + 3, center_pos_mode='protein', max_graph_nodes=max_nodes,
                                 use_session=True)
        for _ in range(3):
            s.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            s.step()
        torch.cuda.synchronize()
        ms = (time.perf_counter() - t0) / steps * 1e3
        n_all, n_dirty, levels = s.session.row_counts()
        return {'ms_per_step': ms, 'value': n / ms, 'layer0_rows': n_dirty, 'receptive_field_levels': levels}
    packed = workloads.pack_samples([pocket], n, sizes)
    for spread in (1.0, 2.0, 3.0, 4.0):
        gen = torch.Generator(device='cpu').manual_seed(2021)
        lpos, lv = workloads.init_ligand(packed, generator=gen, spread=spread)
        out[f'cloud_{spread:.0f}A'] = run(packed, lpos, lv, pocket.num_atoms + max(sizes))
    dpath = os.path.join(ROOT, 'tests', 'golden', 'ligand_1h36_docked.npz')
    if os.path.exists(dpath):
        with np.load(dpath) as z:
            dock = torch.from_numpy(z['pos'].astype(np.float32))
        na = dock.shape[0]
        packed_d = workloads.pack_samples([pocket], n, [na] * n)
        gen = torch.Generator(device='cpu').manual_seed(2021)
        lpos = dock.repeat(n, 1) + 0.25 * torch.randn(n * na, 3, generator=gen)
        _, lv = workloads.init_ligand(packed_d, generator=gen)
        out['docked_pose'] = run(packed_d, lpos, lv, pocket.num_atoms + na)
        out['docked_pose']['ligand_atoms'] = na
    out['steps'] = steps
    return out


def seeded_state_dict(model, seed=2021):
    """Random-init weights of the reference architecture (no checkpoint ships with the reference)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, p in model.named_parameters():
        if not p.requires_grad:
            continue
        if k.endswith('net.1.weight'):
            sd[k] = 1.0 + 0.2 * torch.randn(p.shape, generator=g)
        elif k.endswith('net.1.bias'):
            sd[k] = 0.1 * torch.randn(p.shape, generator=g)
        else:
            fan_in = p.shape[-1] if p.dim() > 1 else p.shape[0]
            sd[k] = (torch.rand(p.shape, generator=g) * 2 - 1) / (fan_in ** 0.5)
    return sd


CPU_THREADS = 16       # fastest of 8 / 16 / 32 / 64 / 128 torch threads on the 128-core MI355X host at 4 samples (1.2 / 1.6 / 3.5 / 7.5 s per
                       # step at 32 / 64 / 128: the small per-layer ops do not scale past one socket's worth of cores)


def cpu_baseline(pocket, sizes, samples=None, steps=2, extrapolate=True, budget_s=20.0):
    """Oracle restatement on the host cores over a bounded sample of the same workload AT ITS REAL BATCH: all the pocket's samples in one
    packed batch (C2: 100 samples, 60,708 nodes), one timed reverse step and a second one if the first took under `budget_s` seconds
    (about 20-60 s of CPU work; the warm-up runs on 4 samples).  The per-step figure is therefore not extrapolated across batch sizes;
    only the 1000-step run length is.  extrapolate=False: the sample IS a whole configuration (BASELINE config 1: 4 samples x 100 steps)
    and is reported as run."""
    from oracle import restatement as R
    from oracle import weights
    sd = weights.make_state_dict(2021)
    if samples is None:
        samples = len(sizes)
    b = workloads.pack_samples(pocket, samples, sizes[:samples])
    g = torch.Generator().manual_seed(0)
    lpos, lv = workloads.init_ligand(b, generator=g)
    nl = lpos.shape[0]
    noises = torch.randn(steps + 1, nl, 3, generator=g)
    unis = torch.rand(steps + 1, nl, 13, generator=g)
    prev = torch.get_num_threads()
    host_cores = os.cpu_count() or 1
    cores = max(1, min(CPU_THREADS, host_cores))
    torch.set_num_threads(cores)
    try:
        args = (sd, None, b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                b.ligand_element_batch)
        if extrapolate:               # warm-up on a small batch of the same pocket (thread pool, allocator)
            wb = workloads.pack_samples(pocket, 4, sizes[:4])
            wpos, wv = workloads.init_ligand(wb, generator=torch.Generator().manual_seed(1))
            wn = wpos.shape[0]
            R.sample_diffusion(sd, None, wb.protein_pos, wb.protein_atom_feature.float(), wb.protein_element_batch, wpos, wv,
                               wb.ligand_element_batch, num_steps=1, noises=torch.zeros(2, wn, 3), uniforms=torch.full((2, wn, 13), 0.5))
            t0 = time.time()
            R.sample_diffusion(*args, num_steps=1, noises=noises, uniforms=unis)
            first = time.time() - t0
            timed, total = 1, first
            if first < budget_s and steps > 1:
                t0 = time.time()
                R.sample_diffusion(*args, num_steps=steps - 1, noises=noises, uniforms=unis)
                total += time.time() - t0
                timed = steps
            steps = timed
            sec_per_step = total / timed
        else:
            R.sample_diffusion(*args, num_steps=1, noises=noises, uniforms=unis)            # warm-up
            t0 = time.time()
            R.sample_diffusion(*args, num_steps=steps, noises=noises, uniforms=unis)
            sec_per_step = (time.time() - t0) / steps
    finally:
        torch.set_num_threads(prev)
    res = {'value': samples / (1000.0 * sec_per_step), 'unit': 'ligands/s', 'cores': cores, 'cores_of_host': f'{cores} of {host_cores}',
           'kind': 'port',
           'sample': f'oracle/restatement.py (torch CPU fp32, {cores} of the host\'s {host_cores} cores = the fastest thread count on the MI355X host), '
                     f'the same pocket at the workload\'s real batch: {samples} samples in one packed batch ({b.protein_pos.shape[0] + nl} nodes) x '
                     f'{steps} reverse step(s), {sec_per_step:.2f} s/step; only the run length (1000 steps) is extrapolated'}
    if not extrapolate:
        res['sample'] = (f'BASELINE config 1 in full: oracle/restatement.py (torch CPU fp32, {cores} threads), 1h36 pocket x {samples} '
                         f'samples x {steps} steps run to completion in {sec_per_step * steps:.1f} s ({sec_per_step:.2f} s/step); value = '
                         f'the same rate expressed per 1000-step ligand')
        res['config1_wall_s'] = sec_per_step * steps
    # the REAL reference cannot run on the GPU box (/root/reference is not there): its own timing, taken in the build container
    # by tools/cpu_reference.py on BASELINE config 1 in full, is quoted beside the port's figure, and what the port's figure on THIS host
    # would read for the real reference at the ratio measured there (same host, same inputs) is stated under a name that says so
    ref_path = os.path.join(ROOT, 'profiles', 'r03_cpu_reference_c1.json')
    if os.path.exists(ref_path):
        with open(ref_path) as f:
            rj = json.load(f)
        ratio = rj.get('port_same_inputs', {}).get('reference_over_port')
        res['reference_cpu'] = {'source': 'profiles/r03_cpu_reference_c1.json (tools/cpu_reference.py, build container)',
                                'config': rj['config'], 'host': rj['host'], 'threads': rj['threads'], 'wall_s': rj['wall_s'],
                                's_per_step': rj['s_per_step'], 'ligands_per_s': rj['ligands_per_s_per_1000_step_ligand'],
                                'reference_over_port_same_host': ratio}
        if ratio:
            res['estimated_reference_ligands_per_s'] = res['value'] / ratio
            res['estimated_reference_note'] = (f'value / {ratio:.2f}: the REAL reference (models/molopt_score_model.py:633-703) took {ratio:.2f} x the '
                                               'port\'s time on the same host and inputs (build container, BASELINE config 1 in full); the reference tree '
                                               'does not exist on the GPU box, so this is an estimate, not a measurement')
        res['sample'] += (f"; the REAL reference on the build container's {rj['threads']} threads: config 1 in full in "
                          f"{rj['wall_s']:.0f} s ({rj['s_per_step']:.2f} s/step)")
    return res


# ---- whole-step FLOP bookkeeping (2 flop per MAC; LayerNorm / activations / exp excluded) --------
GEMM128 = 2 * 128 * 128                              # one 128 x 128 Linear on one row
FIRST_LAYER = 2 * 32 * 128 * 20                      # radial/type first layer of one dst row (32 edges)
H2X_ROW = KEY_PASS_FLOP_EXECUTED + FIRST_LAYER + 2 * 32 * 128 * 16     # key half + xv first layer + xv = W2xv z (16 heads)
GATE_ROW = FIRST_LAYER + 2 * 32 * 128                # edge_pred_layer: 20 -> 128 per edge, 128 -> 1
HEAD_ROW = GEMM128 + 2 * 128 * 13
F_ALG_PER_NODE = 39.35e6                             # canonical, all stages on all edges
F_REDUCED_PER_NODE = 24.5e6                          # same, h2x on ligand dst rows only (section 8d, last bullet)


def executed_flops_per_step(n_nodes, n_lig, n_layers, session_rows, key_row=KEY_PASS_FLOP_EXECUTED, val_row=KEY_PASS_FLOP_EXECUTED,
                            chunks=1):
    """FLOPs the launched kernels execute in one denoiser step, from the rows every launch processes (the same row lists
    run_backbone in csrc/plan.cpp walks).  Stateless forward: every layer runs on every row.  General graphs: `key_row` / `val_row` =
    the attention passes' FLOPs per dst row (all its chunks), `chunks` = chunks per row for the gate and the h2x stage; the node
    projections do not depend on the fan-in."""
    N, Nl, L = n_nodes, n_lig, n_layers
    if session_rows is None:
        x2h_rows = [N] * L
        proj_rows = [N] * L
        hop1 = N
        gate_rows, layer0_proj = N, N
    else:
        levels = session_rows['receptive_field_levels']
        lvl = lambda k: levels[k - 1] if 1 <= k <= len(levels) else N         # receptive-field level k (1-based) or all rows
        x2h_rows, proj_rows = [session_rows['layer0_rows']], [Nl]             # layer 0: dirty rows; projections of ligand rows
        for l in range(1, L):
            e = L - 1 - l
            rows = lvl(e + 1)
            if l == 1 and e + 1 > len(levels) and session_rows.get('layer1_rows') is not None:
                rows = session_rows['layer1_rows']
            x2h_rows.append(rows)
            proj_rows.append(lvl(e + 2))
        hop1 = lvl(1)
        gate_rows = session_rows['layer0_rows']
    f = gate_rows * GATE_ROW * chunks + Nl * HEAD_ROW
    for l in range(L):
        f += proj_rows[l] * 6 * GEMM128                    # k_i, k_j, v_i, v_j, q.net.0, q.net.3
        f += x2h_rows[l] * (key_row + val_row)             # key pass + value pass
        f += hop1 * 2 * GEMM128 + Nl * 4 * GEMM128         # h2x stage: k_j, v_j on the hop rows; k_i, v_i, q.net.0/3 on ligand rows
        f += Nl * H2X_ROW * chunks
    return float(f)


def build_model(dev, args=None):
    cfg = dict(MODEL_CONFIG)
    if args is not None:
        cfg.update(knn=args.knn, cutoff_mode=args.cutoff_mode, r=args.radius, max_num_neighbors=args.cap)
    model = ScorePosNet3D(cfg, workloads.PROTEIN_FEATURE_DIM, workloads.NUM_LIGAND_CLASSES)
    model.load_state_dict(seeded_state_dict(model), strict=False)
    model = model.to(dev).eval()
    if args is not None and args.fp32_node_gemms:
        model.set_native_option('node_proj_split', 0)
    for kv in (args.option if args is not None else []):
        name, value = kv.split('=')
        model.set_native_option(name, int(value))
    return model


def split_error_table():
    """The committed error table of the bf16 x 3 switches (the newest profiles/*_split_error_table.txt)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, 'profiles', '*_split_error_table.txt')))
    return os.path.relpath(files[-1], ROOT) if files else 'tests/test_gpu_graph_modes.py::test_model_options_live_in_the_handle'


def stateless_floor(model, batch, lpos, lv, max_nodes, steps=10, warmup=3):
    """The geometry-independent floor: the same batch through the stateless td_model_forward at every step (no sampling
    session: no static-protein caching, no row pruning -- every layer on every row)."""
    s0 = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos, lv,
                              batch.ligand_element_batch, num_steps=steps + warmup, center_pos_mode='protein',
                              max_graph_nodes=max_nodes, use_session=False)
    for _ in range(warmup):
        s0.step()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    for _ in range(steps):
        s0.step()
    torch.cuda.synchronize()
    return (time.perf_counter() - t1) / steps * 1e3


def graph_desc(args):
    if args.cutoff_mode == 'radius':
        return f'radius {args.radius} A, fan-out cap {args.cap}'
    return f'{args.cutoff_mode} k = {args.knn}'


def full_run(model, pocket, sizes, dev, steps=1000):
    """One complete `sample_diffusion_ligand` call (the reference driver's unit of work, scripts/sample_diffusion.py:31-116):
    every sample of the pocket in one batch, all `steps` reverse steps, wall-clocked end to end (batch construction, the
    session set-up, the loop, the single trajectory D2H copy, un-batching to float64 numpy lists); then the same trajectory
    once more through the stepping interface with HIP events around its first and last 10 steps."""
    from targetdiff_amd import sampling
    n = len(sizes)
    torch.manual_seed(2021)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    out = sampling.sample_diffusion_ligand(model, pocket, n, batch_size=n, device=dev, num_steps=steps, ligand_num_atoms=sizes)
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    finite = all(np.isfinite(p).all() for p in out[0])
    # first / last step times along the same trajectory
    torch.manual_seed(2021)
    pdev = workloads.DevicePocket(pocket, dev)
    batch = workloads.pack_samples_device(pdev, n, sizes)
    lpos, lv = workloads.init_ligand(batch)
    sampler = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos, lv,
                                   batch.ligand_element_batch, num_steps=steps, center_pos_mode='protein',
                                   max_graph_nodes=pocket.num_atoms + max(sizes))
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
    w = min(10, steps // 2)
    for k in range(steps):
        if k == 0:
            ev[0].record()
        if k == w:
            ev[1].record()
        if k == steps - w:
            ev[2].record()
        sampler.step()
    ev[3].record()
    torch.cuda.synchronize()
    n_all, dirty, levels = sampler.session.row_counts()
    return {'call': f'sample_diffusion_ligand(num_samples={n}, batch_size={n}, num_steps={steps})', 'samples': n, 'steps': steps,
            'wall_s': wall, 'ligands_per_s': n / wall * (1000.0 / steps), 'driver_time_list_s': out[6],
            'first_steps_ms': ev[0].elapsed_time(ev[1]) / w, 'last_steps_ms': ev[2].elapsed_time(ev[3]) / w,
            'final_positions_finite': bool(finite),
            'final_ligand_coordinate_std_A': float(np.mean([p.std(axis=0).mean() for p in out[0]])),
            'last_step_session_rows': {'nodes': n_all, 'layer0_rows': dirty, 'receptive_field_levels': levels}}


def run_c4(args, model, dev, rank, world, fence):
    """BASELINE config 4: the test-set job of scripts/batch_sample_diffusion.sh.  Total work is fixed (100 pockets x 100
    samples), pocket i goes to rank i % world; a "step" advances every pocket of the rank by one reverse step."""
    import torch.distributed as dist
    pockets, spp, sizes, desc = make_workload('c4', rank)
    mine = workloads.partition_pockets(len(pockets), world, rank)
    total = args.warmup + args.steps
    samplers, nodes = [], 0
    bs = args.batch_size if args.batch_size and args.batch_size > 0 else spp       # scripts/batch_sample_diffusion.sh:2 runs BATCH_SIZE=50
    for i in mine:
        pdev = workloads.DevicePocket(pockets[i], dev)
        # the pocket's samples in batches of `bs`, one after the other as scripts/sample_diffusion.py:38-41 runs them (a session each)
        for b0 in range(0, spp, bs):
            nb = min(bs, spp - b0)
            bsizes = sizes[b0:b0 + nb]
            batch = workloads.pack_samples_device(pdev, nb, bsizes)
            gen = torch.Generator(device='cpu').manual_seed(2021 + i + 7919 * b0)
            lpos, lv = workloads.init_ligand(workloads.pack_samples(pockets[i], nb, bsizes), generator=gen, spread=args.ligand_spread)
            samplers.append(model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch,
                                                 lpos.to(dev), lv.to(dev), batch.ligand_element_batch, num_steps=total,
                                                 center_pos_mode='protein', max_graph_nodes=pockets[i].num_atoms + max(bsizes),
                                                 use_session=not args.no_session))
            nodes += int(batch.protein_pos.shape[0] + lpos.shape[0])
    for sm in samplers:
        for _ in range(args.warmup):
            sm.step()
    fence()
    t0 = time.perf_counter()
    for sm in samplers:
        for _ in range(args.steps):
            sm.step()
    torch.cuda.synchronize()
    mine_elapsed = time.perf_counter() - t0
    fence()
    elapsed = time.perf_counter() - t0
    per_rank = [mine_elapsed]
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
        tall = [torch.zeros(1, dtype=torch.float64, device=dev) for _ in range(world)]
        dist.all_gather(tall, torch.tensor([mine_elapsed], dtype=torch.float64, device=dev))
        per_rank = [float(t.item()) for t in tall]
    sec_per_step = elapsed / args.steps
    return {
        'metric': METRIC, 'value': len(pockets) * spp / (1000.0 * sec_per_step), 'unit': 'ligands/s', 'n_gpus': world,
        'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': sec_per_step * 1e3, 'higher_is_better': True,
        'scaling': 'strong', 'vs_baseline': None, 'dtype': 'f32',
        'data': 'synthetic (seeded random weights of the reference architecture; synthetic pockets; k-NN rule parity-unpinned upstream)',
        'config': {'workload': desc, 'batch_size': bs, 'batches_per_pocket': (spp + bs - 1) // bs, 'ligand_spread': args.ligand_spread, 'pockets_total': len(pockets),
                   'pockets_this_rank': len(mine), 'nodes_rank0': nodes,
                   'parallelism': f'pocket i -> rank i % {world} (no data-path collective)'},
        'load_balance': {'per_rank_seconds': per_rank, 'max_over_mean': max(per_rank) / (sum(per_rank) / len(per_rank))},
        # what RCCL / the launcher actually gave this job: world size, and per rank its device ordinal, PCI address, own ms per step
        # (fails loudly when two ranks share a device)
        'ranks': launch.rank_census(dev, mine_elapsed / args.steps, {'pockets': len(mine), 'nodes': nodes}),
        'roofline': None}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=3)
    ap.add_argument('--workload', default='c2', choices=['c1', 'c2', 'c3', 'c4', 'c5'])
    ap.add_argument('--batch-size', type=int, default=0, help='c4: samples per batch of a pocket (0 = all 100 in one batch; scripts/batch_sample_diffusion.sh '
                    'runs BATCH_SIZE=50, the signature of sample_diffusion_ligand defaults to 16); the batches of a pocket run one after the other')
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--cpu-full', action='store_true', help='time the CPU baseline on BASELINE config 1 in full (4 samples x 100 '
                    'steps, ~2-5 min of host time) instead of the bounded sample')
    ap.add_argument('--no-full-run', action='store_true', help='skip the complete 1000-step sample_diffusion_ligand call (c2, N = 1)')
    ap.add_argument('--profile-all', action='store_true', help='time every kernel class, print a breakdown to stderr')
    ap.add_argument('--ligand-spread', type=float, default=LIGAND_SPREAD,
                    help='per-coordinate std (A) of the ligand cloud the timed steps start from; 1.0 = the sampler\'s '
                         'initial state N(0, I) (see --initial-state)')
    ap.add_argument('--initial-state', action='store_true',
                    help='after the timed region, also time 10 steps from the sampler\'s initial state N(0, I)')
    ap.add_argument('--no-sweep', action='store_true', help='skip the geometry sweep (c2, N = 1): the session\'s step time at ligand clouds of 1 / 2 / '
                    '3 / 4 A and at the docked pose of the reference\'s example ligand')
    ap.add_argument('--no-graph', action='store_true', help='issue the launches of a step one by one instead of replaying the captured hipGraph')
    ap.add_argument('--no-session', action='store_true', help='stateless td_model_forward per step (no static-protein caching)')
    ap.add_argument('--no-stateless', action='store_true', help='skip the 10 stateless steps reported as stateless_ms_per_step')
    ap.add_argument('--knn', type=int, default=32, help='fan-in of the k-NN / hybrid graph (C5 sweep: 16, 32, 48, 64)')
    ap.add_argument('--cutoff-mode', default='knn', choices=['knn', 'hybrid', 'radius'])
    ap.add_argument('--radius', type=float, default=6.0, help='cut-off (A) of --cutoff-mode radius')
    ap.add_argument('--cap', type=int, default=32, help='fan-out cap of --cutoff-mode radius (C5 sweep)')
    ap.add_argument('--option', action='append', default=[], metavar='NAME=VALUE', help='td_model_set_option (experiments)')
    ap.add_argument('--fp32-node-gemms', action='store_true', help='node-side GEMMs on fp32 MFMA instead of the exact bf16 x 3 split')
    args = ap.parse_args()
    # `--gpus N` without a launcher: become the launcher (N ranks of this same command line), exit with their status
    launch.self_spawn_if_needed(args.gpus)

    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    # under a launcher (torchrun or the self-spawn) the process group is always set up -- also for a world of one rank, so
    # that a single-GPU box exercises the same RCCL initialisation, barrier and reduction as the N-rank job
    distributed = world > 1 or launch.under_launcher()
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs a HIP device (MI355X); there is no CPU path')
    torch.cuda.set_device(local_rank)
    dev = torch.device('cuda', local_rank)
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        with launch.stdout_to_stderr():          # RCCL's version banner goes to stdout: keep rank 0's stdout to the one JSON line
            dist.init_process_group('nccl', device_id=dev)          # RCCL over xGMI; rendezvous + timing max only
            dist.barrier()
            torch.cuda.synchronize()

    def fence():
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()

    model = build_model(dev, args)
    default_graph = args.cutoff_mode == 'knn' and args.knn <= 32      # 32-slot rows: the fast path and the caching session
    if args.workload == 'c4':
        out = run_c4(args, model, dev, rank, world, fence)
        if rank == 0:
            print(json.dumps(out))
        if distributed:
            dist.barrier()
            dist.destroy_process_group()
        return

    pockets, spp, sizes, desc = make_workload(args.workload, rank)
    batch = workloads.pack_samples(pockets, spp, sizes).to(dev)
    gen = torch.Generator(device='cpu').manual_seed(2021 + rank)
    lpos, lv = workloads.init_ligand(workloads.pack_samples(pockets, spp, sizes), generator=gen, spread=args.ligand_spread)
    lpos, lv = lpos.to(dev), lv.to(dev)
    n_nodes = int(batch.protein_pos.shape[0] + lpos.shape[0])
    n_lig = int(lpos.shape[0])
    max_nodes = max(p.num_atoms for p in pockets) + max(sizes)
    args.prof_steps = max(1, min(args.steps, 10))
    total = args.warmup + args.steps + args.prof_steps
    if total > 1000:
        raise SystemExit('warmup + steps + the profiled steps must be <= 1000 (one sampling run)')
    sampler = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch,
                                   lpos, lv, batch.ligand_element_batch, num_steps=total, center_pos_mode='protein',
                                   max_graph_nodes=max_nodes, use_session=not args.no_session, use_graph=False if args.no_graph else None)
    for _ in range(args.warmup):
        sampler.step()

    # the timed region: K steps, nothing else on the stream (no per-class timers)
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sampler.step()
    fence()
    elapsed = time.perf_counter() - t0
    step_launch = ('one hipGraph replay per step (td_session_step)' if sampler.session is not None and sampler.session.last_step_was_graph()
                   else 'launch by launch')
    # roofline leg: the next steps of the same run with HIP events around the x2h key / value launches, recorded on the
    # launch stream (td_profile_begin / td_profile_end); their launch times feed `roofline`, not `value`
    classes = capi.PROFILE_CLASSES if args.profile_all else ('x2h_k', 'x2h_v')
    prof_steps = args.prof_steps
    capi.profile_begin(classes)
    for _ in range(prof_steps):
        sampler.step()
    torch.cuda.synchronize()
    prof = capi.profile_end()
    my_elapsed = elapsed
    if distributed:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    # what RCCL / the launcher actually gave this job (gathered on every rank; fails loudly when two ranks share a device)
    census = launch.rank_census(dev, my_elapsed / args.steps, {'nodes': n_nodes})
    sec_per_step = elapsed / args.steps
    graphs = len(pockets) * spp
    value = world * graphs / (1000.0 * sec_per_step)

    # dst rows per key / value launch, averaged over the 9 layers of the last step: with the session the first layer only
    # recomputes the rows a ligand atom touches and the last layer only the ligand atoms' 1-hop neighbourhood
    n_layers = MODEL_CONFIG['num_layers']
    rows_per_launch = float(n_nodes)
    session_rows = None
    if sampler.session is not None:
        n_all, n_dirty, levels = sampler.session.row_counts()
        tail = levels[:n_layers - 1]          # level k + 1 = rows the layer k from the end updates
        n_fwd = sampler.session.forward_reach_rows()
        full_layers = n_layers - 1 - len(tail) - (1 if n_fwd is not None else 0)
        rows_per_launch = (n_dirty + (n_fwd or 0) + full_layers * n_all + sum(tail)) / n_layers
        session_rows = {'nodes': n_all, 'layer0_rows': n_dirty, 'layer1_rows': n_fwd, 'receptive_field_levels': levels}

    # general graphs: a dst row is ceil(fan-in / 32) chunks of 32 slots; the first layer and the head-shaped products run per
    # chunk, the 128 x 128 output product of the value pass once per row (the key pass rebuilds U_i per chunk)
    fan_in = args.cap if args.cutoff_mode == 'radius' else args.knn
    cpn = (fan_in + 31) // 32
    flop_key = cpn * KEY_PASS_FLOP_EXECUTED
    # (the value pass runs a chunk whose second 16-slot block is all padding -- slots 48 .. 63 at k = 48 -- on its first block only)
    last = fan_in - 32 * (cpn - 1)
    slots_val = 32 * (cpn - 1) + (16 if (cpn > 1 and last <= 16 and args.cutoff_mode == 'knn') else 32)
    flop_val = 2 * (slots_val * 128 * 20 + slots_val * 128 * 16) + 2 * 128 * 128
    split = bool(model._native(dev).get_option('edge_key_split'))
    l1_f16 = split and bool(model._native(dev).get_option('edge_first_layer_f16'))       # x2h passes: first layer on f16 piece pairs
    first_exec = FIRST_LAYER_FLOP_F16_EXECUTED if l1_f16 else FIRST_LAYER_FLOP_BF16_EXECUTED
    # second layer on f16 piece pairs: the value pass on every graph, the key pass on rows of one chunk (the default graph; the protein rows of
    # `hybrid` / k < 32 / capped-radius graphs, which run the default graph's kernels through the chunk index) and, beside the f16 first layer,
    # on the chunk walk
    l2_f16 = bool(model._native(dev).get_option('edge_second_layer_f16')) and split
    l2_key = l2_f16 and (default_graph or cpn == 1 or l1_f16)      # (the chunk-walking key pass: f16 logits only beside the f16 first layer)

    def pass_roofline(cls, kernel, traffic_file):
        p = prof[cls]
        if not p['launches']:
            return None
        ms = p['ms'] / p['launches']
        # 32-slot rows with fewer than 32 edges (k < 32, radius cap < 32): the per-edge terms count the edges that exist
        per_row = (2 * (fan_in * 128 * 20 + 128 * 128 + fan_in * 128 * 16) if fan_in <= 32
                   else (flop_key if cls == 'x2h_k' else flop_val))
        achieved = per_row * rows_per_launch / (ms * 1e-3) / 1e12
        # matrix-pipe bound of the kernel as built (algorithmic TFLOP/s): the first layer's share at the bf16 peak when it runs
        # on piece triples, the second layer's at the 16-bit peak when it runs on f16 piece pairs, the rest at the fp32 peak
        chunks = cpn if fan_in > 32 else 1
        first_alg = 2 * min(fan_in, 32) * 128 * 20 * chunks
        split_here = split
        l2_here = l2_key if cls == 'x2h_k' else l2_f16
        if split_here and l2_here:
            # + the second layer (logits / alpha^T z: 2 * 32 * 128 * 16 algorithmic FLOPs per chunk) as three f16 piece products at the 16-bit peak
            second_alg = 2 * (min(fan_in, 32) if fan_in <= 32 else (32 * cpn if cls == 'x2h_k' else slots_val)) * 128 * 16
            t_min = (chunks * (first_exec + 3 * 2 * 32 * 128 * 16) / PEAK_BF16_MFMA_TFLOPS
                     + (per_row - first_alg - second_alg) / PEAK_FP32_MFMA_TFLOPS)
            bound = per_row / t_min
        elif split_here:
            t_min = cpn * first_exec / PEAK_BF16_MFMA_TFLOPS + (per_row - first_alg) / PEAK_FP32_MFMA_TFLOPS
            bound = per_row / t_min
        else:
            bound = PEAK_FP32_MFMA_TFLOPS
        # HBM traffic of the same kernel: PMC counters cannot be read in-process, so this is the figure of the COMMITTED
        # profile of the same command (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes, tools/pmc_collect.sh;
        # FETCH_SIZE doubled per MI355X_MICROARCH.md section HBM) -- static, see traffic_source
        traffic, source = None, None
        if args.workload != 'c2':
            traffic_file = traffic_file.replace('.json', f'_{args.workload}.json')
        tpath = os.path.join(ROOT, 'profiles', traffic_file)
        if os.path.exists(tpath) and sampler.session is not None and default_graph:
            with open(tpath) as f:
                tj = json.load(f)
            same = tj.get('build_tag') == capi.build_tag()
            if same:
                traffic = (2.0 * tj['fetch_kb'] + tj['write_kb']) * 1024.0
                source = (f'profiles/{traffic_file}: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command on this very library '
                          f'build ({tj["build_tag"]}; PMC counters cannot be read in-process), tools/refresh_headline.sh')
            else:
                source = (f'profiles/{traffic_file} describes library build {tj.get("build_tag")}, this run is build {capi.build_tag()}: '
                          'stale, not reported (tools/refresh_headline.sh re-collects the PMC passes)')
        return {'bound': 'mfma', 'kernel': kernel, 'rows_per_launch': rows_per_launch, 'session_rows': session_rows, 'achieved': achieved,
                'peak': PEAK_FP32_MFMA_TFLOPS, 'unit': 'TFLOP/s', 'frac': achieved / PEAK_FP32_MFMA_TFLOPS,
                'matrix_bound_as_built': bound, 'frac_of_matrix_bound_as_built': achieved / bound,
                'first_layer': (('f16 piece pairs (22 bits per operand), K-packed: 2 x v_mfma_f32_16x16x32_f16 per tile' if l1_f16
                                 else 'bf16 x 3 piece triples, K-packed: 4 x v_mfma_f32_16x16x32_bf16 per tile') if split_here else 'fp32 (v_mfma_f32_16x16x4_f32)'),
                'second_layer': ('f16 piece pairs (22 bits per operand), 3 x v_mfma_f32_16x16x32_f16 per tile; the per-row 128 x 128 product on the vector unit'
                                 if (split_here and l2_here) else 'fp32 (v_mfma_f32_16x16x4_f32); the per-row 128 x 128 product on the vector unit'),
                'traffic': traffic, 'traffic_source': source, 'launch_ms': ms, 'launches': p['launches'],
                # secondary bound: HBM bytes actually moved per launch (PMC) against the 8 TB/s roofline
                'hbm_gbs': (traffic / (ms * 1e-3) / 1e9) if traffic else None,
                'hbm_frac': (traffic / (ms * 1e-3) / 1e9 / PEAK_HBM_GBS) if traffic else None,
                'achieved_canonical_formulation': KEY_PASS_FLOP_CANONICAL * rows_per_launch / (ms * 1e-3) / 1e12,
                'flop_per_node_executed': KEY_PASS_FLOP_EXECUTED, 'flop_per_node_canonical': KEY_PASS_FLOP_CANONICAL,
                'share_of_step': (p['ms'] / prof_steps) / (sec_per_step * 1e3), 'profiled_steps': prof_steps}

    if default_graph:
        vk, kk = (('edge_value16t_kernel<L2, false, FL> (12 waves)', 'edge_key16_kernel<false, 12, 0, 0, true, L2, FL>') if split
                  else ('edge_value16_kernel<false>', 'edge_key16_kernel<false, 16, 0, 0, false>'))
    elif split and cpn == 1:      # one chunk per protein row: the default graph's kernels through the chunk index, the ligand rows in a second launch
        vk = 'edge_value16t_kernel<L2, true, FL> (12 waves; protein rows) + edge_value16_kernel<true, true, false, L2, FL> (chunk-walking; ligand rows)'
        kk = 'edge_key16_kernel<false, 12, 0, 2, true, L2, FL> (protein rows) + edge_key16_kernel<false, 12, 0, 1, true, L2, FL> (chunk-walking; ligand rows)'
    else:
        vk = 'edge_value16_kernel<true, true, false, L2, FL> (chunk-walking)' if split else 'edge_value16_kernel<false, true> (chunk-walking)'
        kk = 'edge_key16_kernel<false, 12, 0, 1, true, L2, FL> (chunk-walking)' if split else 'edge_key16_kernel<false, 16, 0, 1, false>'
    roofline = pass_roofline('x2h_v', vk + ' (x2h value pass)', 'traffic_x2h_value.json')
    if roofline is not None:
        roofline['key_pass'] = pass_roofline('x2h_k', kk + ' (x2h key pass)', 'traffic_x2h_key.json')
    # whole step: FLOPs the launched kernels execute (from the row lists) against the fp32 peak, next to SURVEY 8d's algorithmic
    # figures (which count work the session provably does not need to do: fractions above 1 there only say the eliminations are real)
    if default_graph:
        f_exec = executed_flops_per_step(n_nodes, n_lig, n_layers, session_rows)
    else:                       # per-chunk work scales with the chunks per row (uniform for k-NN / radius; hybrid: the protein rows' count)
        f_exec = executed_flops_per_step(n_nodes, n_lig, n_layers, None, flop_key, flop_val, cpn)
    if fan_in < 32:             # fewer edges per row: scale the per-edge share (about 80 % of a row's FLOPs) -- an estimate
        f_exec *= 0.2 + 0.8 * fan_in / 32.0
    whole_step = {'executed_flop': f_exec, 'executed_tflops': f_exec / sec_per_step / 1e12,
                  'executed_frac_of_fp32_peak': f_exec / sec_per_step / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                  'f_alg_flop': F_ALG_PER_NODE * n_nodes, 'f_alg_tflops': F_ALG_PER_NODE * n_nodes / sec_per_step / 1e12,
                  'f_reduced_flop': F_REDUCED_PER_NODE * n_nodes,
                  'f_reduced_frac_of_fp32_peak': F_REDUCED_PER_NODE * n_nodes / sec_per_step / 1e12 / PEAK_FP32_MFMA_TFLOPS,
                  'note': 'executed = 2 flop/MAC of every Linear / attention product the launched kernels run on the rows they run '
                          'on; f_alg / f_reduced = SURVEY.md section 8d (all rows, every layer)'}
    if roofline is not None:
        roofline['whole_step'] = whole_step
    out = {
        'metric': METRIC, 'value': value, 'unit': 'ligands/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': sec_per_step * 1e3, 'higher_is_better': True, 'scaling': 'weak',
        'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic (seeded random weights of the reference architecture; '
        + ('real 1h36 pocket geometry' if args.workload in ('c1', 'c2') else 'synthetic pockets')
        + '; k-NN rule (d2 association, ties -> lower index) is the project\'s: torch_cluster is not in the reference tree, '
          'parity-unpinned upstream; arithmetic fp32 throughout, the node-side 128 x 128 GEMMs '
        + ('on fp32 MFMA' if args.fp32_node_gemms else 'on an exact 3-way bf16 split of both fp32 operands with fp32 accumulation '
           f'(fp32-equivalent: errors against the reference golden unchanged, {split_error_table()})')
        + (('; the 21-wide radial/type first layer of the attention kernels (x2h passes, h2x stage) on f16 piece pairs: weights (scaled by a power of two per MLP) and '
            'inputs carried to 22 bits (the exact bf16 x 3 form stays selectable: edge_first_layer_f16 = 0; the edge gate uses it)'
            if l1_f16 else '; the 21-wide radial/type first layer of the x2h attention passes on the same kind of split') if split else '')
        + ('; their per-edge second-layer products (logits, alpha^T z) on f16 piece pairs: operands carried to 22 bits, within one to two fp32 '
           'roundings of the fp32 products (tests/test_gpu_weight_regimes.py holds every form to the same goldens)' if l2_f16 else '') + ')',
        'config': {'workload': desc, 'graph': graph_desc(args), 'ligand_spread': args.ligand_spread, 'nodes_per_gpu': n_nodes,
                   'edges_per_gpu': (32 if default_graph else fan_in) * n_nodes, 'graphs_per_gpu': graphs,
                   'node_gemms': 'fp32 MFMA' if args.fp32_node_gemms else 'exact bf16 x 3 operand split, fp32 accumulate',
                   'edge_first_layer': (('f16 piece pairs of both operands (22 significant bits each), fp32 accumulate (x2h passes and h2x stage; edge gate: exact bf16 x 3 split)'
                                         if l1_f16 else 'exact bf16 x 3 operand split, fp32 accumulate') if split else 'fp32 MFMA'),
                   'edge_second_layer': (('f16 piece pairs of both operands (22 significant bits each), fp32 accumulate'
                                          + ('' if l2_key else '; the chunk-walking key pass: fp32 MFMA')) if l2_f16 else 'fp32 MFMA'),
                   'step_launch': step_launch, 'build_tag': capi.build_tag(),
                   'parallelism': f'pocket-sharded x{world} (no data-path collective)'},
        'roofline': roofline,
        'ranks': census,
    }
    # the sampler's own initial state (ligand cloud N(0, I)): cheaper steps, reported beside the headline number
    if args.initial_state and args.ligand_spread != 1.0 and sampler.session is not None:
        gen0 = torch.Generator(device='cpu').manual_seed(2021 + rank)
        lpos0, lv0 = workloads.init_ligand(workloads.pack_samples(pockets, spp, sizes), generator=gen0, spread=1.0)
        s0 = model.begin_sampling(batch.protein_pos, batch.protein_atom_feature.float(), batch.protein_element_batch,
                                  lpos0.to(dev), lv0.to(dev), batch.ligand_element_batch, num_steps=13,
                                  center_pos_mode='protein', max_graph_nodes=max_nodes, use_session=True)
        for _ in range(3):
            s0.step()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for _ in range(10):
            s0.step()
        torch.cuda.synchronize()
        ms0 = (time.perf_counter() - t1) / 10 * 1e3
        n0, d0, lv0c = s0.session.row_counts()
        out['initial_state'] = {'ligand_spread': 1.0, 'ms_per_step': ms0, 'value': world * graphs / ms0, 'steps': 10,
                                'session_rows': {'nodes': n0, 'layer0_rows': d0, 'receptive_field_levels': lv0c}}
    if world == 1 and sampler.session is not None and not args.no_stateless:
        ms_floor = stateless_floor(model, batch, lpos, lv, max_nodes)
        out['stateless_ms_per_step'] = ms_floor
        out['stateless_value'] = graphs / ms_floor
        out['stateless_note'] = ('10 steps of the same batch without the sampling session (td_model_forward per step: every layer '
                                 'on every row, no caching) -- the geometry-independent floor of `value`')
    del sampler
    if rank == 0:
        if args.profile_all:
            for k, v in prof.items():
                if v['launches']:
                    print(f'  {k:10s} {v["ms"] / prof_steps:9.3f} ms/step  ({v["launches"] // prof_steps} launches/step)',
                          file=sys.stderr)
        # a complete run the driver's own clock can witness: outside the timed region, the 20-step line above is unchanged
        if world == 1 and args.workload == 'c2' and not args.no_sweep and not args.no_session and default_graph:
            out['geometry_sweep'] = geometry_sweep(model, pockets[0], sizes, dev)
        if world == 1 and args.workload == 'c2' and not args.no_full_run and not args.no_session and default_graph:
            out['full_run'] = full_run(model, pockets[0], sizes, dev)
        if world == 1 and not args.no_cpu_baseline:
            if args.cpu_full:
                c1p, c1n, c1s, _ = make_workload('c1', 0)
                out['cpu_baseline'] = cpu_baseline(c1p[0], c1s, samples=c1n, steps=100, extrapolate=False)
            else:
                out['cpu_baseline'] = cpu_baseline(pockets[0], sizes)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
