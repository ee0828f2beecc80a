"""Host-side input construction for the denoising hot path: pockets, ragged packing of
pocket x sample graphs, noise initialisation.

Mirrors what the reference driver does before it calls the sampler:

* ``Batch.from_data_list([data.clone()] * n_data, follow_batch=FOLLOW_BATCH)``
  (scripts/sample_diffusion.py:42, datasets/pl_data.py:7): concatenate every per-atom tensor and
  emit ``<key>_batch`` graph-id vectors -> :func:`pack_samples`.
* ligand initialisation (scripts/sample_diffusion.py:60-70): centroid + N(0, I) positions and a
  Gumbel-arg-max of uniform logits for the atom types -> :func:`init_ligand`.
* the fixed-column PDB ATOM parser + protein atom featuriser used by scripts/sample_for_pocket.py:18-31
  (utils/data.py:64-95,99-118; utils/transforms.py:115-132) -> :func:`pocket_from_pdb`.
* synthetic pockets of BASELINE.json configs 3-5 (recipe: SURVEY.md section 8d) -> :func:`synthetic_pocket`.

Everything here is host-side numpy/torch plumbing; the device work lives in ``csrc/``.
"""
from __future__ import annotations

from dataclasses import dataclass

import numpy as np
import torch

# utils/transforms.py:118 (H, C, N, O, S, Se) and utils/data.py:24-33 (amino-acid order).
PROTEIN_ELEMENTS = (1, 6, 7, 8, 16, 34)
AA_NAMES = ('ALA', 'CYS', 'ASP', 'GLU', 'PHE', 'GLY', 'HIS', 'ILE', 'LYS', 'LEU', 'MET', 'ASN', 'PRO',
            'GLN', 'ARG', 'SER', 'THR', 'VAL', 'TRP', 'TYR')
BACKBONE_NAMES = ('CA', 'C', 'N', 'O')
_SYMBOL_TO_Z = {'H': 1, 'C': 6, 'N': 7, 'O': 8, 'S': 16, 'Se': 34, 'P': 15, 'F': 9, 'Cl': 17}
PROTEIN_FEATURE_DIM = len(PROTEIN_ELEMENTS) + len(AA_NAMES) + 1     # 27
NUM_LIGAND_CLASSES = 13                                              # utils/transforms.py:48-62 (add_aromatic)


@dataclass
class Pocket:
    """One protein pocket: ``pos`` [n,3] float32, ``feat`` [n,27] int64 one-hot blocks."""
    pos: np.ndarray
    feat: np.ndarray
    name: str = ''

    @property
    def num_atoms(self) -> int:
        return int(self.pos.shape[0])


def featurize_protein(element: np.ndarray, aa_type: np.ndarray, is_backbone: np.ndarray) -> np.ndarray:
    """FeaturizeProteinAtom.__call__ (utils/transforms.py:126-132): [element==Z | one_hot(aa, 20) | backbone]."""
    elem = (element.reshape(-1, 1) == np.asarray(PROTEIN_ELEMENTS).reshape(1, -1)).astype(np.int64)
    aa = np.zeros((len(element), len(AA_NAMES)), dtype=np.int64)
    aa[np.arange(len(element)), aa_type] = 1
    return np.concatenate([elem, aa, is_backbone.reshape(-1, 1).astype(np.int64)], axis=1)


def pocket_from_pdb(path_or_block: str, name: str = '') -> Pocket:
    """Fixed-column ATOM parser (utils/data.py:64-95) + featuriser.  HETATM/hydrogens are kept exactly as
    the reference keeps them (it filters nothing at this level)."""
    block = path_or_block
    if '\n' not in path_or_block:
        with open(path_or_block, 'r') as f:
            block = f.read()
    pos, elem, aa, bb = [], [], [], []
    for line in block.splitlines():
        tag = line[0:6].strip()
        if tag == 'ENDMDL':
            break
        if tag != 'ATOM':
            continue
        sym = line[76:78].strip().capitalize() or line[13:14]
        if sym not in _SYMBOL_TO_Z:
            raise ValueError(f'unsupported element {sym!r} in pocket file')
        pos.append([float(line[30:38]), float(line[38:46]), float(line[46:54])])
        elem.append(_SYMBOL_TO_Z[sym])
        aa.append(AA_NAMES.index(line[17:20].strip()))
        bb.append(line[12:16].strip() in BACKBONE_NAMES)
    if not pos:
        raise ValueError('no ATOM records found')     # datasets/pl_pair_dataset.py:106 asserts non-empty protein
    return Pocket(np.asarray(pos, np.float32),
                  featurize_protein(np.asarray(elem), np.asarray(aa), np.asarray(bb)), name)


def synthetic_pocket(seed: int, n_atoms: int = 300, r_in: float = 4.0, r_out: float = 14.0,
                     min_sep: float = 1.2) -> Pocket:
    """atoms uniform in the shell r_in < r < r_out with a minimum
    separation; element ~ Cat(C .65, N .15, O .19, S .01), residue ~ U{0..19}, backbone ~ Bern(.5)."""
    rng = np.random.RandomState(seed)
    pts = np.zeros((0, 3), dtype=np.float64)
    while len(pts) < n_atoms:
        cand = rng.uniform(-r_out, r_out, size=(4 * n_atoms, 3))
        r = np.linalg.norm(cand, axis=1)
        cand = cand[(r > r_in) & (r < r_out)]
        for c in cand:
            if len(pts) == 0 or np.min(np.linalg.norm(pts - c, axis=1)) >= min_sep:
                pts = np.vstack([pts, c[None]])
                if len(pts) == n_atoms:
                    break
    elem = rng.choice([6, 7, 8, 16], size=n_atoms, p=[.65, .15, .19, .01])
    aa = rng.randint(0, 20, size=n_atoms)
    bb = rng.rand(n_atoms) < 0.5
    shift = rng.uniform(-20, 20, size=(1, 3))        # pockets are not centred in the reference data
    return Pocket((pts + shift).astype(np.float32), featurize_protein(elem, aa, bb), f'synthetic_{seed}')


def synthetic_test_set(n_pockets: int = 100, seed: int = 4000):
    """Stand-in for the CrossDocked test split of BASELINE config 4 (the dataset is absent):
    ``n_pockets`` synthetic pockets with 250-600 atoms (seeded), shell radius grown with the atom count so that the
    density stays at the real pocket's ~0.03 atoms / A^3."""
    rng = np.random.RandomState(seed)
    out = []
    for i in range(n_pockets):
        n = int(rng.randint(250, 601))
        r_out = float((4.0 ** 3 + n / (4.0 / 3.0 * np.pi * 0.027)) ** (1.0 / 3.0))
        out.append(synthetic_pocket(seed + 1 + i, n, 4.0, r_out))
    return out


@dataclass
class PackedBatch:
    """The ragged pack the sampler consumes (``Batch.from_data_list`` attributes the driver reads)."""
    protein_pos: torch.Tensor          # [N_p, 3] f32
    protein_atom_feature: torch.Tensor  # [N_p, 27] i64
    protein_element_batch: torch.Tensor  # [N_p] i64  (graph id)
    ligand_element_batch: torch.Tensor   # [N_l] i64
    ligand_num_atoms: list
    num_graphs: int

    def to(self, device):
        return PackedBatch(self.protein_pos.to(device), self.protein_atom_feature.to(device),
                           self.protein_element_batch.to(device), self.ligand_element_batch.to(device),
                           self.ligand_num_atoms, self.num_graphs)


def pack_samples(pockets, samples_per_pocket, ligand_num_atoms) -> PackedBatch:
    """Replicate each pocket ``samples_per_pocket`` times and pack all graphs into one ragged batch
    (scripts/sample_diffusion.py:42,48-50).  ``ligand_num_atoms``: one size per graph, in graph order."""
    if isinstance(pockets, Pocket):
        pockets = [pockets]
    pos, feat, bp, sizes = [], [], [], []
    g = 0
    for p in pockets:
        for _ in range(samples_per_pocket):
            pos.append(torch.from_numpy(p.pos))
            feat.append(torch.from_numpy(p.feat))
            bp.append(torch.full((p.num_atoms,), g, dtype=torch.long))
            g += 1
    ligand_num_atoms = [int(v) for v in ligand_num_atoms]
    assert len(ligand_num_atoms) == g, (len(ligand_num_atoms), g)
    bl = torch.repeat_interleave(torch.arange(g), torch.tensor(ligand_num_atoms))
    return PackedBatch(torch.cat(pos), torch.cat(feat), torch.cat(bp), bl, ligand_num_atoms, g)


class DevicePocket:
    """One pocket resident on the device: the only host -> device copy the driver makes per pocket."""

    def __init__(self, pocket: Pocket, device):
        self.num_atoms = pocket.num_atoms
        self.pos = torch.from_numpy(pocket.pos).to(device)
        self.feat = torch.from_numpy(pocket.feat).to(device)
        self.name = pocket.name


def pack_samples_device(pocket: DevicePocket, n: int, ligand_num_atoms) -> PackedBatch:
    """``Batch.from_data_list([data.clone() for _ in range(n)], follow_batch=FOLLOW_BATCH).to(device)``
    (scripts/sample_diffusion.py:42) without the n host-side clones and the n-fold H2D copy: the replicas are laid out
    on the device from the single resident pocket (same attribute values as :func:`pack_samples`)."""
    dev = pocket.pos.device
    sizes = [int(v) for v in ligand_num_atoms]
    assert len(sizes) == n, (len(sizes), n)
    ids = torch.arange(n, device=dev)
    bp = ids.repeat_interleave(pocket.num_atoms)
    bl = ids.repeat_interleave(torch.tensor(sizes, device=dev), output_size=sum(sizes))
    return PackedBatch(pocket.pos.repeat(n, 1), pocket.feat.repeat(n, 1), bp, bl, sizes, n)


def init_ligand(batch: PackedBatch, num_classes: int = NUM_LIGAND_CLASSES, generator=None, spread: float = 1.0, draw=None,
                types: bool = True):
    """scripts/sample_diffusion.py:60-70: positions = protein centroid + N(0, I); types = arg-max of Gumbel
    noise over uniform logits (models/molopt_score_model.py:160-166).  `spread` != 1 scales the noise (benchmarks use
    it to emulate the geometry of a later, spread-out ligand; the reference's initial state is spread = 1).
    ``draw(name, like)`` may supply the Gaussian ('noise') / uniform draws (parity tests)."""
    dev = batch.protein_pos.device
    B = batch.num_graphs
    if dev.type == 'cuda':
        # deterministic per-graph reduction in the HIP library (an index_add_ / scatter_mean on the GPU sums with atomics
        # and makes the starting positions differ in the last bit from run to run)
        from . import capi
        cen = capi.protein_centroids(batch.protein_pos, capi.graph_ptr(batch.protein_element_batch.contiguous(), B))
    else:
        s = torch.zeros(B, 3, device=dev).index_add_(0, batch.protein_element_batch, batch.protein_pos)
        c = torch.bincount(batch.protein_element_batch, minlength=B).clamp(min=1).unsqueeze(-1).float()
        cen = s / c
    center = cen[batch.ligand_element_batch]
    n = center.shape[0]
    if draw is not None:
        pos = center + spread * draw('noise', center)
    else:
        pos = center + spread * torch.randn(n, 3, generator=generator, device=dev)
    if not types:               # pos_only: the driver takes the types from the data and draws no uniforms (:66-67)
        return pos, None
    if draw is not None:
        u = draw('uniform', torch.empty(n, num_classes, dtype=torch.float32, device=dev))
    else:
        u = torch.rand(n, num_classes, generator=generator, device=dev)
    v = (-torch.log(-torch.log(u + 1e-30) + 1e-30)).argmax(dim=-1)
    return pos, v


def partition_pockets(num_pockets: int, world_size: int, rank: int, start_idx: int = 0, costs=None):
    """The reference's only parallelism: pocket i goes to worker i % NODE_ALL
    (scripts/batch_sample_diffusion.sh:15-20) -- the default.

    ``costs`` (opt-in; one number per pocket, e.g. its node count = protein atoms + expected ligand atoms): size-balanced
    assignment instead -- longest-processing-time-first: pockets in descending cost order, each to the currently least
    loaded rank (ties: lower rank).  Deterministic, every rank computes the same table, still no communication."""
    if costs is None:
        return [i for i in range(start_idx, num_pockets) if i % world_size == rank]
    return lpt_assignment(costs, world_size, start_idx)[rank]


def lpt_assignment(costs, world_size: int, start_idx: int = 0):
    """-> per rank, the ascending list of pocket indices under longest-processing-time-first."""
    order = sorted(range(start_idx, len(costs)), key=lambda i: (-float(costs[i]), i))
    load = [0.0] * world_size
    out = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (load[k], k))
        out[r].append(i)
        load[r] += float(costs[i])
    return [sorted(v) for v in out]


def predicted_imbalance(costs, world_size: int, balanced: bool = False, start_idx: int = 0):
    """max / mean of the per-rank cost sums under the round-robin (default) or the LPT assignment"""
    if balanced:
        parts = lpt_assignment(costs, world_size, start_idx)
    else:
        parts = [[i for i in range(start_idx, len(costs)) if i % world_size == r] for r in range(world_size)]
    sums = [sum(float(costs[i]) for i in p) for p in parts]
    mean = sum(sums) / max(len(sums), 1)
    return max(sums) / mean if mean > 0 else 1.0
