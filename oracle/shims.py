"""Pure-torch stand-ins for the third-party ops on the reference hot path (TEST INFRASTRUCTURE).

The reference pins pytorch-scatter 2.1.0, pytorch-cluster 1.6.0 and pyg 2.2.0
(/root/reference/environment.yaml:185,196,199); none of them is installed here and their sources are
not under /root/reference, so the arithmetic below restates their *published* semantics
(SURVEY.md Appendix B).  Call sites pinned on:

* ``knn_graph``       models/uni_transformer.py:280, models/egnn.py:99, models/common.py:199
* ``scatter_softmax`` models/uni_transformer.py:73,135
* ``scatter_sum``     models/uni_transformer.py:78,139, models/egnn.py:53,61
* ``scatter_mean``    models/molopt_score_model.py:115, scripts/sample_diffusion.py:61

The kNN tie/rounding rule is unpinned upstream (CUDA kernel vs CPU KD-tree disagree on ties); the rule
adopted for this project -- and followed bit-for-bit by the HIP kernel -- is:

    d2(i, j) = (dx*dx + dy*dy) + dz*dz     in fp32, no FMA contraction, dx = x_j - x_i
    neighbours of i = the k smallest (d2, j) pairs, j != i, j in the same graph, ascending.
"""
from __future__ import annotations

import sys
import types

import torch


# --------------------------------------------------------------------------- torch_scatter
def _broadcast_index(index: torch.Tensor, src: torch.Tensor, dim: int) -> torch.Tensor:
    if dim < 0:
        dim = src.dim() + dim
    if index.dim() == 1:
        for _ in range(dim):
            index = index.unsqueeze(0)
    while index.dim() < src.dim():
        index = index.unsqueeze(-1)
    return index.expand(src.size())


def scatter_sum(src, index, dim=-1, out=None, dim_size=None):
    index = _broadcast_index(index, src, dim)
    if out is None:
        size = list(src.size())
        if dim_size is not None:
            size[dim] = dim_size
        elif index.numel() == 0:
            size[dim] = 0
        else:
            size[dim] = int(index.max()) + 1
        out = torch.zeros(size, dtype=src.dtype, device=src.device)
    return out.scatter_add_(dim, index, src)


def scatter_add(src, index, dim=-1, out=None, dim_size=None):
    return scatter_sum(src, index, dim, out, dim_size)


def scatter_mean(src, index, dim=-1, out=None, dim_size=None):
    out = scatter_sum(src, index, dim, out, dim_size)
    dim_size = out.size(dim)
    index_dim = dim
    if index_dim < 0:
        index_dim = index_dim + src.dim()
    if index.dim() <= index_dim:
        index_dim = index.dim() - 1
    ones = torch.ones(index.size(), dtype=src.dtype, device=src.device)
    count = scatter_sum(ones, index, index_dim, None, dim_size)
    count[count < 1] = 1
    count = _broadcast_index(count, out, dim) if count.dim() == 1 else count
    if out.is_floating_point():
        out.true_divide_(count)
    else:
        out.div_(count, rounding_mode='floor')
    return out


def scatter_max(src, index, dim=-1, dim_size=None):
    index_b = _broadcast_index(index, src, dim)
    size = list(src.size())
    size[dim] = dim_size if dim_size is not None else int(index.max()) + 1
    out = torch.full(size, float('-inf'), dtype=src.dtype, device=src.device)
    out = out.scatter_reduce(dim, index_b, src, reduce='amax', include_self=True)
    return out


def scatter_softmax(src, index, dim=-1, dim_size=None):
    if not torch.is_floating_point(src):
        raise ValueError('scatter_softmax needs floating point input')
    index_b = _broadcast_index(index, src, dim)
    max_per = scatter_max(src, index, dim, dim_size)
    rec = (src - max_per.gather(dim, index_b)).exp_()
    sum_per = scatter_sum(rec, index_b, dim, dim_size=dim_size)
    return rec.div(sum_per.gather(dim, index_b))


def scatter(src, index, dim=-1, out=None, dim_size=None, reduce='sum'):
    if reduce in ('sum', 'add'):
        return scatter_sum(src, index, dim, out, dim_size)
    if reduce == 'mean':
        return scatter_mean(src, index, dim, out, dim_size)
    raise NotImplementedError(reduce)


# --------------------------------------------------------------------------- torch_geometric.nn
def knn_neighbours(x: torch.Tensor, k: int, batch: torch.Tensor | None = None) -> torch.Tensor:
    """Dense neighbour table [N, k] (int64, -1 padded) under the project's kNN rule (module docstring)."""
    N = x.size(0)
    if batch is None:
        batch = torch.zeros(N, dtype=torch.long, device=x.device)
    assert bool((batch[1:] >= batch[:-1]).all()), 'batch must be sorted'
    out = torch.full((N, k), -1, dtype=torch.long, device=x.device)
    if N == 0:
        return out
    counts = torch.bincount(batch)
    start = 0
    for n in counts.tolist():
        if n == 0:
            continue
        xs = x[start:start + n].to(torch.float32)
        dx = xs[None, :, 0] - xs[:, None, 0]
        dy = xs[None, :, 1] - xs[:, None, 1]
        dz = xs[None, :, 2] - xs[:, None, 2]
        d2 = (dx * dx + dy * dy) + dz * dz            # separate mul/add kernels: no FMA contraction
        d2.fill_diagonal_(float('inf'))
        kk = min(k, n - 1)
        if kk > 0:
            # stable sort => ties resolved towards the lower index
            order = torch.sort(d2, dim=1, stable=True).indices[:, :kk]
            out[start:start + n, :kk] = order + start
        start += n
    return out


def _d2_matrix(xq: torch.Tensor, xc: torch.Tensor) -> torch.Tensor:
    """[nq, nc] squared distances under the project's rule: (dx*dx + dy*dy) + dz*dz in fp32, separate mul / add."""
    xq, xc = xq.to(torch.float32), xc.to(torch.float32)
    dx = xc[None, :, 0] - xq[:, None, 0]
    dy = xc[None, :, 1] - xq[:, None, 1]
    dz = xc[None, :, 2] - xq[:, None, 2]
    return (dx * dx + dy * dy) + dz * dz


def hybrid_neighbours(x: torch.Tensor, k: int, mask_ligand: torch.Tensor, batch: torch.Tensor) -> torch.Tensor:
    """Dense in-neighbour table [N, width] (-1 padded) of ``batch_hybrid_edge_connection(x, k, mask_ligand, batch,
    add_p_index=True)`` (models/common.py:165-212, dispatched at models/uni_transformer.py:281-283):

    * a protein atom: its k nearest nodes of the graph (protein or ligand) -- the knn_graph rows restricted to protein dst
      (:197-204);
    * a ligand atom: every other ligand atom of the graph (:166-171), then its k nearest protein atoms (:173-181).

    The reference picks the protein neighbours with ``torch.topk`` on ``torch.norm`` distances (tie order and the
    rounding of the norm are unspecified); the project's rule is the kNN one: ascending (d2, index) with the fp32 d2 above.
    The reference's edge ORDER is irrelevant to its scatter ops, so parity is on the neighbour SETS (tests)."""
    N = x.size(0)
    mask_ligand = mask_ligand.bool()
    rows = [None] * N
    counts = torch.bincount(batch)
    start = 0
    for n in counts.tolist():
        idx = torch.arange(start, start + n)
        lig = idx[mask_ligand[idx]]
        prot = idx[~mask_ligand[idx]]
        if len(prot):
            d2 = _d2_matrix(x[prot], x[idx])
            d2[torch.arange(len(prot)), prot - start] = float('inf')          # prot is a prefix of idx in compose order
            kk = min(k, n - 1)
            order = torch.sort(d2, dim=1, stable=True).indices[:, :kk] + start
            for r, i in enumerate(prot.tolist()):
                rows[i] = order[r].tolist()
        if len(lig):
            kp = min(k, len(prot))
            near = torch.sort(_d2_matrix(x[lig], x[prot]), dim=1, stable=True).indices[:, :kp] if kp else None
            for r, i in enumerate(lig.tolist()):
                rows[i] = [j for j in lig.tolist() if j != i] + (prot[near[r]].tolist() if kp else [])
        start += n
    width = max((len(r) for r in rows), default=0)
    out = torch.full((N, max(width, 1)), -1, dtype=torch.long)
    for i, r in enumerate(rows):
        out[i, :len(r)] = torch.tensor(r, dtype=torch.long)
    return out


def radius_neighbours(x: torch.Tensor, r: float, batch: torch.Tensor, max_num_neighbors: int = 32) -> torch.Tensor:
    """Dense in-neighbour table [N, max_num_neighbors] (-1 padded) of the radius graph with a fan-out cap.

    The reference cannot run this mode (models/uni_transformer.py:278 reads ``self.r``, never assigned), so there is no
    reference behaviour; the rule is this project's, after the published semantics of torch_cluster's ``radius_graph``
    (first hits in index order, not nearest first): the first ``max_num_neighbors`` nodes j != i of the same graph, in
    ascending index order, with d2(i, j) < r * r (fp32 d2 as above, r * r rounded to fp32, strict).  (torch_cluster 1.6.0
    searches max_num_neighbors + 1 hits including i itself and drops i afterwards, which returns one edge more for nodes
    whose first hits all precede them; that artefact is not reproduced.)"""
    N = x.size(0)
    out = torch.full((N, max_num_neighbors), -1, dtype=torch.long)
    r2 = torch.tensor(r, dtype=torch.float32) * torch.tensor(r, dtype=torch.float32)
    counts = torch.bincount(batch)
    start = 0
    for n in counts.tolist():
        d2 = _d2_matrix(x[start:start + n], x[start:start + n])
        hit = d2 < r2
        hit.fill_diagonal_(False)
        for li in range(n):
            js = torch.nonzero(hit[li]).squeeze(1)[:max_num_neighbors] + start
            out[start + li, :len(js)] = js
        start += n
    return out


def knn_graph(x, k, batch=None, loop=False, flow='source_to_target', cosine=False, num_workers=1):
    """edge_index [2, E]: row 0 = source (neighbour), row 1 = target (query); grouped by target ascending,
    ascending distance inside a group (torch_geometric.nn.knn_graph, flow='source_to_target')."""
    assert not loop and not cosine and flow == 'source_to_target'
    nbr = knn_neighbours(x, k, batch)
    dst = torch.arange(x.size(0), device=x.device).view(-1, 1).expand_as(nbr)
    keep = nbr >= 0
    return torch.stack([nbr[keep], dst[keep]], dim=0)


def radius_graph(x, r, batch=None, loop=False, max_num_neighbors=32, flow='source_to_target', num_workers=1):
    raise NotImplementedError('radius_graph is dead code in the reference (models/uni_transformer.py:278 '
                              'reads self.r which is never assigned)')


# --------------------------------------------------------------------------- torch_geometric.data / .transforms
class Data:
    """Minimal stand-in for ``torch_geometric.data.Data`` as the sampling driver uses it (datasets/pl_data.py:10-36,
    scripts/sample_diffusion.py:42): an attribute bag of tensors with ``clone()``."""

    def __init__(self, **kwargs):
        for k, v in kwargs.items():
            setattr(self, k, v)

    def keys(self):
        return [k for k in self.__dict__ if not k.startswith('_')]

    def __getitem__(self, k):
        return getattr(self, k)

    def __setitem__(self, k, v):
        setattr(self, k, v)

    def clone(self):
        return type(self)(**{k: (v.clone() if torch.is_tensor(v) else v) for k, v in self.__dict__.items()})

    def to(self, device):
        for k, v in list(self.__dict__.items()):
            if torch.is_tensor(v):
                setattr(self, k, v.to(device))
        return self


class Batch(Data):
    """``Batch.from_data_list(list, follow_batch=K)``: every tensor attribute concatenated along
    dim 0 (``*_index`` keys along the last dim, shifted by the running ligand atom count, datasets/pl_data.py:32-36) and
    ``<key>_batch`` = graph id per row for each key in K."""

    @classmethod
    def from_data_list(cls, data_list, follow_batch=()):
        out = cls()
        keys = data_list[0].keys()
        for k in keys:
            vals = [d[k] for d in data_list]
            if not torch.is_tensor(vals[0]):
                setattr(out, k, vals)
                continue
            if k.endswith('index'):
                shift, parts = 0, []
                for d, v in zip(data_list, vals):
                    parts.append(v + shift)
                    shift += int(d['ligand_element'].size(0)) if hasattr(d, 'ligand_element') else 0
                setattr(out, k, torch.cat(parts, dim=-1))
            else:
                setattr(out, k, torch.cat(vals, dim=0))
            if k in follow_batch:
                setattr(out, k + '_batch', torch.repeat_interleave(
                    torch.arange(len(vals)), torch.tensor([int(v.size(0)) for v in vals])))
        out.num_graphs = len(data_list)
        return out


class Compose:
    def __init__(self, transforms):
        self.transforms = list(transforms)

    def __call__(self, data):
        for t in self.transforms:
            data = t(data)
        return data


# --------------------------------------------------------------------------- easydict
class EasyDict(dict):
    """Attribute dict used by the reference config loader (utils/misc.py:23-25)."""

    def __init__(self, d=None, **kwargs):
        super().__init__()
        d = dict(d or {})
        d.update(kwargs)
        for key, val in d.items():
            self[key] = val

    def __setitem__(self, key, val):
        if isinstance(val, dict) and not isinstance(val, EasyDict):
            val = EasyDict(val)
        super().__setitem__(key, val)

    def __getattr__(self, key):
        try:
            return self[key]
        except KeyError as exc:
            raise AttributeError(key) from exc

    __setattr__ = __setitem__


def install() -> None:
    """Register the shims under the third-party module names the reference imports."""
    if 'torch_scatter' not in sys.modules:
        m = types.ModuleType('torch_scatter')
        for name in ('scatter_sum', 'scatter_add', 'scatter_mean', 'scatter_softmax', 'scatter_max', 'scatter'):
            setattr(m, name, globals()[name])
        sys.modules['torch_scatter'] = m
    if 'torch_geometric' not in sys.modules:
        tg = types.ModuleType('torch_geometric')
        tgnn = types.ModuleType('torch_geometric.nn')
        tgnn.knn_graph = knn_graph
        tgnn.radius_graph = radius_graph
        tg.nn = tgnn
        tgdata = types.ModuleType('torch_geometric.data')
        tgdata.Data, tgdata.Batch = Data, Batch
        tgtr = types.ModuleType('torch_geometric.transforms')
        tgtr.Compose = Compose
        tg.data, tg.transforms = tgdata, tgtr
        sys.modules['torch_geometric'] = tg
        sys.modules['torch_geometric.nn'] = tgnn
        sys.modules['torch_geometric.data'] = tgdata
        sys.modules['torch_geometric.transforms'] = tgtr
    if 'easydict' not in sys.modules:
        ed = types.ModuleType('easydict')
        ed.EasyDict = EasyDict
        sys.modules['easydict'] = ed
