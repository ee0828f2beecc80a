"""Mirror of the reference's EGNN refine net (models/egnn.py): same class names, constructor arguments and
``state_dict`` keys; the arithmetic runs in libtargetdiff_hip.so (td_egnn_forward).  Built for the configuration
``get_refine_net('egnn', config)`` produces (models/molopt_score_model.py:34-42): num_r_gaussian = 1, kNN graph, SiLU,
no LayerNorm, hidden 128, 4 edge types, k = 32, coordinate update on.  Anything else raises.

The reference cannot reach this net through ScorePosNet3D (its forward passes ``fix_x`` to the refine net, which EGNN does
not accept -- SURVEY.md section 0), so, as there, it is a standalone ``forward(h, x, mask_ligand, batch)`` module.
"""
from __future__ import annotations

import torch
from torch import nn

from . import capi

_FIXED_OFFSETS = [0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10]


class _Holder(nn.Module):
    def forward(self, *a, **k):
        raise RuntimeError('parameter holder: the arithmetic runs in libtargetdiff_hip.so')


class _Mlp2(_Holder):
    """models/common.py:60-80 with num_layer=2, norm=False: keys net.0 / net.2."""

    def __init__(self, in_dim, out_dim, hidden_dim, act_last):
        super().__init__()
        layers = [nn.Linear(in_dim, hidden_dim), nn.SiLU(), nn.Linear(hidden_dim, out_dim)]
        if act_last:
            layers.append(nn.SiLU())
        self.net = nn.Sequential(*layers)


class EnBaseLayer(_Holder):
    """models/egnn.py:10-35 (parameters only)."""

    def __init__(self, hidden_dim, edge_feat_dim, num_r_gaussian, update_x=True, act_fn='silu', norm=False):
        super().__init__()
        if num_r_gaussian != 1 or not update_x or act_fn != 'silu' or norm:
            raise NotImplementedError('EnBaseLayer: built for num_r_gaussian=1, update_x=True, act_fn="silu", norm=False')
        self.hidden_dim, self.edge_feat_dim, self.num_r_gaussian = hidden_dim, edge_feat_dim, num_r_gaussian
        self.update_x, self.act_fn, self.norm = update_x, act_fn, norm
        self.edge_mlp = _Mlp2(2 * hidden_dim + edge_feat_dim + num_r_gaussian, hidden_dim, hidden_dim, act_last=True)
        self.edge_inf = nn.Sequential(nn.Linear(hidden_dim, 1), nn.Sigmoid())
        layer = nn.Linear(hidden_dim, 1, bias=False)
        torch.nn.init.xavier_uniform_(layer.weight, gain=0.001)                     # models/egnn.py:30
        self.x_mlp = nn.Sequential(nn.Linear(hidden_dim, hidden_dim), nn.SiLU(), layer, nn.Tanh())
        self.node_mlp = _Mlp2(2 * hidden_dim, hidden_dim, hidden_dim, act_last=False)


class EGNN(nn.Module):
    """models/egnn.py:67-133: ``forward(h, x, mask_ligand, batch, return_all=False)`` -> {'x', 'h'[, 'all_x', 'all_h']}."""

    def __init__(self, num_layers, hidden_dim, edge_feat_dim, num_r_gaussian, k=32, cutoff=10.0, cutoff_mode='knn',
                 update_x=True, act_fn='silu', norm=False):
        super().__init__()
        if cutoff_mode != 'knn' or (hidden_dim, edge_feat_dim, k) != (128, 4, 32):
            raise NotImplementedError(f'EGNN: built for cutoff_mode="knn", hidden 128, edge_feat_dim 4, k 32 '
                                      f'(got {cutoff_mode!r}, {hidden_dim}, {edge_feat_dim}, {k})')
        self.num_layers, self.hidden_dim, self.edge_feat_dim, self.num_r_gaussian = num_layers, hidden_dim, edge_feat_dim, num_r_gaussian
        self.update_x, self.act_fn, self.norm, self.k, self.cutoff, self.cutoff_mode = update_x, act_fn, norm, k, cutoff, cutoff_mode
        self.distance_expansion = _Holder()
        self.distance_expansion.register_buffer('offset', torch.tensor(_FIXED_OFFSETS, dtype=torch.float32))   # :83 (unused)
        self.net = nn.ModuleList([EnBaseLayer(hidden_dim, edge_feat_dim, num_r_gaussian, update_x=update_x, act_fn=act_fn,
                                              norm=norm) for _ in range(num_layers)])
        self._native = None
        self._native_key = None

    def _fingerprint(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _get_native(self, device):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError(f'targetdiff_amd runs on HIP devices only (got {device}); there is no CPU path')
        key = self._fingerprint(device)
        if self._native is None or key != self._native_key:
            self._native = capi.NativeEgnn(self.num_layers, self.state_dict(), self.hidden_dim, self.edge_feat_dim, self.k,
                                           device=device)
            self._native_key = key
        return self._native

    @torch.no_grad()
    def forward(self, h, x, mask_ligand, batch, return_all=False):
        native = self._get_native(h.device)
        from .models import _check_sorted
        _check_sorted(batch)
        B = int(batch.max().item()) + 1 if batch.numel() else 0
        node_ptr = capi.graph_ptr(batch.contiguous(), B)
        h, x = h.contiguous().float(), x.contiguous().float()
        out_h, out_x, all_h, all_x = native.forward(h, x, mask_ligand, node_ptr, return_all=return_all)
        outputs = {'x': out_x, 'h': out_h}
        if return_all:
            outputs.update({'all_x': [x] + list(all_x.unbind(0)), 'all_h': [h] + list(all_h.unbind(0))})
        return outputs
