"""Batching driver of the sampler: mirror of ``sample_diffusion_ligand``
(scripts/sample_diffusion.py:31-116; imported by scripts/sample_for_pocket.py:12) and of the
pocket-level data parallelism of scripts/batch_sample_diffusion.sh:15-20.

Same arguments, same 7-tuple result (lists of numpy arrays, positions as float64), same wall-clock
``time_list`` per sample batch.  ``data`` is duck-typed: anything with ``protein_pos`` [n,3] and
``protein_atom_feature`` [n,27] (a PyG ``ProteinLigandData`` in the reference, a ``workloads.Pocket``
here) -- ``Batch.from_data_list([data.clone()] * n)`` (:42) is replaced by ``workloads.pack_samples``.
"""
from __future__ import annotations

import os
import time

import numpy as np
import torch

from . import workloads


def _as_pocket(data) -> workloads.Pocket:
    if isinstance(data, workloads.Pocket):
        return data
    pos = data.protein_pos
    feat = data.protein_atom_feature
    to_np = lambda t: t.detach().cpu().numpy() if torch.is_tensor(t) else np.asarray(t)
    return workloads.Pocket(to_np(pos).astype(np.float32), to_np(feat).astype(np.int64), getattr(data, 'name', ''))


def _prior_sizes(pocket, n, atom_num_sampler):
    """sample_num_atoms='prior' (:47-50).  The size prior's lookup table lives in the reference's
    utils/evaluation/atom_num_config.py (host-side, out of scope); use it when importable (drop-in inside the
    reference repo) or a caller-supplied sampler ``f(protein_pos_numpy) -> int``."""
    if atom_num_sampler is not None:
        return [int(atom_num_sampler(pocket.pos)) for _ in range(n)]
    try:
        from utils.evaluation import atom_num          # the reference module, if we run inside that repo
    except Exception as exc:
        raise RuntimeError("sample_num_atoms='prior' needs the reference's utils.evaluation.atom_num on the path "
                           'or an explicit atom_num_sampler / ligand_num_atoms') from exc
    size = atom_num.get_space_size(pocket.pos)
    return [int(atom_num.sample_atom_num(size)) for _ in range(n)]


def unbatch_v_traj(ligand_v_traj, n_data, ligand_cum_atoms):
    """scripts/sample_diffusion.py:21-28, vectorised: list over samples of [num_steps, num_atoms_i, ...]."""
    arr = np.stack([v.cpu().numpy() if torch.is_tensor(v) else np.asarray(v) for v in ligand_v_traj])
    return [arr[:, ligand_cum_atoms[k]:ligand_cum_atoms[k + 1]] for k in range(n_data)]


def sample_diffusion_ligand(model, data, num_samples, batch_size=16, device='cuda:0', num_steps=None,
                            pos_only=False, center_pos_mode='protein', sample_num_atoms='prior',
                            atom_num_sampler=None, ligand_num_atoms=None, generator=None, noise_source=None,
                            overlap_batches=False, max_resident_batches=8, use_graph=None):
    """Returns (pred_pos, pred_v, pred_pos_traj, pred_v_traj, pred_v0_traj, pred_vt_traj, time_list).

    Extra keyword arguments (not in the reference signature; all optional): ``atom_num_sampler`` / ``ligand_num_atoms``
    replace the size prior's lookup table, ``generator`` seeds the initial draws, and ``noise_source(batch_index, step,
    name, like)`` injects every Gaussian / uniform draw (``step == -1``: the initial positions / types of :60-70; ``step
    >= 0``: the sampler's per-step draws) -- the parity tests use it to replay the reference's draws.

    ``overlap_batches`` (default False: the reference's order, one batch after the other, torch's global generator consumed
    batch-major exactly as scripts/sample_diffusion.py does): when True, the sample batches of the pocket (independent of each
    other, :40) advance together, each on its own HIP stream.  A small batch (the signature's default batch_size=16 is ~10 k
    nodes) cannot fill the GPU -- its step is a chain of ~50 dependent launches -- so overlapping the chains of several batches
    raises throughput (1.35x at batch_size 16).  In this mode every batch draws from its OWN generator, seeded from torch's global
    generator in batch order when the batch is set up, so the result does not depend on how the batches interleave (it differs from
    the sequential order's sample, equally distributed); injected draws (``noise_source``) give the same bits in both orders.  At
    most ``max_resident_batches`` batches (sessions + trajectories) are alive at a time: the pocket's batches are processed in
    groups of that size.  ``time_list`` then holds, per batch, the wall time of its group divided evenly.  ``use_graph``: see
    ScorePosNet3D.sample_diffusion (None = a captured hipGraph per batch wherever the batch runs on a real stream, i.e. in the
    overlapped mode)."""
    pocket = _as_pocket(data)
    pocket_dev = None
    time_list = []
    num_batch = int(np.ceil(num_samples / batch_size))
    overlap_batches = bool(overlap_batches) and num_batch > 1 and hasattr(model, 'begin_sampling') and torch.device(device).type == 'cuda'
    max_resident_batches = max(1, int(max_resident_batches))
    current_i = 0
    jobs = []                      # overlap_batches: (n_data, sizes, sampler) per sample batch
    parts = None                   # the six result lists accumulated so far
    t_all = time.time()
    for i in range(num_batch):
        n_data = batch_size if i < num_batch - 1 else num_samples - batch_size * (num_batch - 1)
        t1 = time.time()
        if ligand_num_atoms is not None:
            sizes = [int(v) for v in ligand_num_atoms[current_i:current_i + n_data]]
        elif sample_num_atoms == 'prior':
            sizes = _prior_sizes(pocket, n_data, atom_num_sampler)
        elif sample_num_atoms == 'range':
            sizes = list(range(current_i + 1, current_i + n_data + 1))                      # :51-53
        elif sample_num_atoms == 'ref':                                                         # :54-56
            ref_lig = getattr(data, 'ligand_element', None)
            if ref_lig is None:
                ref_lig = getattr(data, 'ligand_pos', None)
            if ref_lig is None:
                raise ValueError("sample_num_atoms='ref' needs the reference ligand on `data` (ligand_element / ligand_pos)")
            sizes = [int(len(ref_lig))] * n_data
        else:
            raise ValueError(sample_num_atoms)
        if pocket_dev is None:
            pocket_dev = workloads.DevicePocket(pocket, device)       # one H2D copy of the pocket for all batches
        batch = workloads.pack_samples_device(pocket_dev, n_data, sizes)                          # :42
        src0 = None if noise_source is None else (lambda name, like, _i=i: noise_source(_i, -1, name, like))
        init_pos, init_v = workloads.init_ligand(batch, model.num_classes, generator=generator, draw=src0,
                                                 types=not pos_only)                               # :60-70
        if pos_only:                                                                              # :66-67
            full = getattr(data, 'ligand_atom_feature_full', None)
            if full is None:
                raise ValueError('pos_only=True takes the atom types from data.ligand_atom_feature_full')
            full = torch.as_tensor(full, dtype=torch.long)
            if any(sz != full.numel() for sz in sizes):
                raise ValueError("pos_only=True needs ligands of the reference size (sample_num_atoms='ref')")
            init_v = full.repeat(n_data).to(device)
        kw = dict(protein_pos=batch.protein_pos, protein_v=batch.protein_atom_feature.float(),
                  batch_protein=batch.protein_element_batch, init_ligand_pos=init_pos, init_ligand_v=init_v,
                  batch_ligand=batch.ligand_element_batch, num_steps=num_steps, pos_only=pos_only,
                  center_pos_mode=center_pos_mode, max_graph_nodes=pocket.num_atoms + max(sizes))
        if use_graph is not None:
            kw['use_graph'] = use_graph
        if noise_source is not None:
            kw['noise_source'] = (lambda st, name, like, _i=i: noise_source(_i, st, name, like))
        if overlap_batches:
            if noise_source is None:
                # the batch's own stream of draws (see the docstring): seeded from the global generator, in batch order
                g_batch = torch.Generator(device=device)
                g_batch.manual_seed(int(torch.randint(0, 2 ** 62, (1,), dtype=torch.int64).item()))
                kw['generator'] = g_batch
            jobs.append((n_data, sizes, model.begin_sampling(
                kw.pop('protein_pos'), kw.pop('protein_v'), kw.pop('batch_protein'), kw.pop('init_ligand_pos'),
                kw.pop('init_ligand_v'), kw.pop('batch_ligand'), **kw)))
            if len(jobs) == max_resident_batches or i == num_batch - 1:
                part = _unbatch(_run_overlapped(jobs, device), pos_only)
                parts = part if parts is None else tuple(a + b for a, b in zip(parts, part))
                time_list += [(time.time() - t_all) / len(jobs)] * len(jobs)
                jobs = []
                t_all = time.time()
        else:
            part = _unbatch([(n_data, sizes, model.sample_diffusion(**kw))], pos_only)      # inside the timed span, as :86-114
            parts = part if parts is None else tuple(a + b for a, b in zip(parts, part))
            time_list.append(time.time() - t1)
        current_i += n_data
    if parts is None:
        parts = ([], [], [], [], [], [])
    return parts + (time_list,)


def _run_overlapped(jobs, device):
    """Advance every batch's sampler by one reverse step per round, each on its own stream."""
    dev = torch.device(device)
    main = torch.cuda.current_stream(dev)
    streams = [torch.cuda.Stream(device=dev) for _ in jobs]
    for st in streams:
        st.wait_stream(main)                   # the samplers were set up on the caller's stream
    pending = True
    while pending:
        pending = False
        for (_, _, sampler), st in zip(jobs, streams):
            if not sampler.done:
                with torch.cuda.stream(st):
                    sampler.step()
                pending = True
    out = []
    for (n_data, sizes, sampler), st in zip(jobs, streams):
        with torch.cuda.stream(st):
            out.append((n_data, sizes, sampler.finish()))          # the trajectory D2H copy synchronises the stream
        main.wait_stream(st)
    return out


def _unbatch(collected, pos_only):
    """scripts/sample_diffusion.py:86-112: per-sample numpy lists (positions as float64) from the packed results."""
    all_pos, all_v, all_pos_traj, all_v_traj, all_v0_traj, all_vt_traj = [], [], [], [], [], []
    for n_data, sizes, r in collected:
        cum = np.cumsum([0] + sizes)
        pos = r['pos'].cpu().numpy().astype(np.float64)
        all_pos += [pos[cum[k]:cum[k + 1]] for k in range(n_data)]                           # :87-90
        pos_traj = np.stack([p.numpy() for p in r['pos_traj']]).astype(np.float64)
        all_pos_traj += [pos_traj[:, cum[k]:cum[k + 1]] for k in range(n_data)]              # :92-99
        v = r['v'].cpu().numpy()
        all_v += [v[cum[k]:cum[k + 1]] for k in range(n_data)]                               # :102-103
        all_v_traj += unbatch_v_traj(r['v_traj'], n_data, cum)
        if not pos_only:                                                                     # :108-112
            all_v0_traj += unbatch_v_traj(r['v0_traj'], n_data, cum)
            all_vt_traj += unbatch_v_traj(r['vt_traj'], n_data, cum)
    return all_pos, all_v, all_pos_traj, all_v_traj, all_v0_traj, all_vt_traj


# ------------------------------------------------------------------------------------------ multi-GPU
def run_sharded(model, pockets, num_samples, rank=0, world_size=1, start_idx=0, result_path=None, skip_existing=True,
                keep_results=None, on_pocket=None, balance=False, **kwargs):
    """Pocket-level data parallelism: pocket i is sampled by rank i % world_size, from ``start_idx`` on
    (scripts/batch_sample_diffusion.sh:13-20).  No data-path collective exists on this path; the caller may gather the
    per-rank result metadata (see ``gather_metadata``).  ``balance=True`` (opt-in): size-balanced assignment by protein
    atom count instead of the round-robin (``workloads.partition_pockets(costs=...)``).

    ``result_path``: write ``result_{i}.pt`` per pocket in the reference layout (scripts/sample_diffusion.py:175-188)
    from a background thread, and -- ``skip_existing`` -- skip pockets whose file is already there (idempotent re-runs;
    the reference resumes by hand through START_IDX).  Returns {pocket index: 7-tuple} for the pockets sampled in this
    call (``keep_results=False`` drops the tuples once written and returns {index: None}: a test set's trajectories are
    gigabytes).  ``on_pocket(index, seconds, skipped)`` is called after every pocket."""
    import time as _time
    from . import results as _results
    keep = (result_path is None) if keep_results is None else keep_results
    out = {}
    writer = _results.AsyncResultWriter() if result_path is not None else None
    try:
        costs = [_as_pocket(p).num_atoms for p in pockets] if balance else None
        for idx in workloads.partition_pockets(len(pockets), world_size, rank, start_idx, costs=costs):
            path = _results.result_file(result_path, idx) if result_path is not None else None
            if path is not None and skip_existing and os.path.exists(path):
                if on_pocket:
                    on_pocket(idx, 0.0, True)
                continue
            t0 = _time.time()
            res = sample_diffusion_ligand(model, pockets[idx], num_samples, **kwargs)
            if writer is not None:
                writer.submit(path, _results.result_dict(pockets[idx], res))
            out[idx] = res if keep else None
            if on_pocket:
                on_pocket(idx, _time.time() - t0, False)
    finally:
        if writer is not None:
            writer.close()
    return out


def gather_metadata(local: dict, group=None):
    """Gather small per-rank python metadata (timings, counts) on every rank.  Works with RCCL ('nccl'
    backend on ROCm) and gloo; returns [local] when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local, group=group)
    return out
