"""Batching driver of the sampler: mirror of ``sample_diffusion_ligand``
(scripts/sample_diffusion.py:31-116; imported by scripts/sample_for_pocket.py:12) and of the
pocket-level data parallelism of scripts/batch_sample_diffusion.sh:15-20.

Same arguments, same 7-tuple result (lists of numpy arrays, positions as float64), same wall-clock
``time_list`` per sample batch.  ``data`` is duck-typed: anything with ``protein_pos`` [n,3] and
``protein_atom_feature`` [n,27] (a PyG ``ProteinLigandData`` in the reference, a ``workloads.Pocket``
here) -- ``Batch.from_data_list([data.clone()] * n)`` (:42) is replaced by ``workloads.pack_samples``.
"""
from __future__ import annotations

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
                            atom_num_sampler=None, ligand_num_atoms=None, generator=None):
    """Returns (pred_pos, pred_v, pred_pos_traj, pred_v_traj, pred_v0_traj, pred_vt_traj, time_list)."""
    pocket = _as_pocket(data)
    all_pos, all_v, all_pos_traj, all_v_traj, all_v0_traj, all_vt_traj, time_list = [], [], [], [], [], [], []
    num_batch = int(np.ceil(num_samples / batch_size))
    current_i = 0
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
        batch = workloads.pack_samples(pocket, n_data, sizes).to(device)
        init_pos, init_v = workloads.init_ligand(batch, model.num_classes, generator=generator)   # :60-70
        if pos_only:                                                                              # :66-67
            full = getattr(data, 'ligand_atom_feature_full', None)
            if full is None:
                raise ValueError('pos_only=True takes the atom types from data.ligand_atom_feature_full')
            full = torch.as_tensor(full, dtype=torch.long)
            if any(sz != full.numel() for sz in sizes):
                raise ValueError("pos_only=True needs ligands of the reference size (sample_num_atoms='ref')")
            init_v = full.repeat(n_data).to(device)
        r = model.sample_diffusion(
            protein_pos=batch.protein_pos, protein_v=batch.protein_atom_feature.float(),
            batch_protein=batch.protein_element_batch, init_ligand_pos=init_pos, init_ligand_v=init_v,
            batch_ligand=batch.ligand_element_batch, num_steps=num_steps, pos_only=pos_only,
            center_pos_mode=center_pos_mode, max_graph_nodes=pocket.num_atoms + max(sizes))
        cum = np.cumsum([0] + sizes)
        pos = r['pos'].cpu().numpy().astype(np.float64)
        all_pos += [pos[cum[k]:cum[k + 1]] for k in range(n_data)]                           # :87-90
        pos_traj = np.stack([p.numpy() for p in r['pos_traj']]).astype(np.float64)
        all_pos_traj += [pos_traj[:, cum[k]:cum[k + 1]] for k in range(n_data)]              # :92-99
        v = r['v'].cpu().numpy()
        all_v += [v[cum[k]:cum[k + 1]] for k in range(n_data)]                               # :102-103
        all_v_traj += unbatch_v_traj(r['v_traj'], n_data, cum)
        all_v0_traj += unbatch_v_traj(r['v0_traj'], n_data, cum)      # empty lists (pos_only) raise here, as in the reference
        all_vt_traj += unbatch_v_traj(r['vt_traj'], n_data, cum)
        time_list.append(time.time() - t1)
        current_i += n_data
    return all_pos, all_v, all_pos_traj, all_v_traj, all_v0_traj, all_vt_traj, time_list


# ------------------------------------------------------------------------------------------ multi-GPU
def run_sharded(model, pockets, num_samples, rank=0, world_size=1, start_idx=0, **kwargs):
    """Pocket-level data parallelism: pocket i is sampled by rank i % world_size
    (scripts/batch_sample_diffusion.sh:15-20).  No data-path collective exists on this path; the caller
    may gather the per-rank result metadata (see ``gather_metadata``)."""
    results = {}
    for idx in workloads.partition_pockets(len(pockets), world_size, rank, start_idx):
        results[idx] = sample_diffusion_ligand(model, pockets[idx], num_samples, **kwargs)
    return results


def gather_metadata(local: dict, group=None):
    """Gather small per-rank python metadata (timings, counts) on every rank.  Works with RCCL ('nccl'
    backend on ROCm) and gloo; returns [local] when torch.distributed is not initialised."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return [local]
    out = [None] * dist.get_world_size(group)
    dist.all_gather_object(out, local, group=group)
    return out
