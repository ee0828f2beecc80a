"""Host-side mirror of the reference's model interface for the denoising path.

Same class names, constructor arguments, ``state_dict`` keys and call signatures as
``models/molopt_score_model.py`` / ``models/uni_transformer.py`` so that
``scripts/sample_diffusion.py`` and ``scripts/sample_for_pocket.py`` run unchanged
 -- but the modules below only *hold* parameters.  All arithmetic of the path runs in
libtargetdiff_hip.so (hand-written HIP for gfx950) through ``capi.NativeModel``; there is no PyTorch or
CPU implementation of the layers in this package, and calling them without the library / a HIP device
raises.

Seams mirrored (reference file:line):
  ScorePosNet3D.__init__            models/molopt_score_model.py:194-311
  ScorePosNet3D.forward             models/molopt_score_model.py:313-368
  ScorePosNet3D.sample_diffusion    models/molopt_score_model.py:633-703
  get_refine_net                    models/molopt_score_model.py:13-45
  UniTransformerO2TwoUpdateGeneral  models/uni_transformer.py:213-328  (forward :301-328)
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn as nn

from . import capi

_FIXED_OFFSETS = (0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10)


def _cfg_get(config, key, default=None):
    if isinstance(config, dict):
        return config.get(key, default)
    return getattr(config, key, default)


def _check_graph_inputs(batch_protein, batch_ligand, ligand_v, num_classes, allow_unsorted=False):
    """Host-side input checks at the points where the reference would raise or re-order; returns ``(protein_unsorted,
    ligand_unsorted)``:

    * ``compose_context`` stable-sorts the concatenated nodes by graph id (models/common.py:126), so unsorted ``batch_*`` vectors
      are legal there.  The HIP path builds CSR offsets from sorted vectors (td_graph_ptr): callers on that seam pass
      ``allow_unsorted=True`` and re-order their inputs with :func:`_stable_order` when a flag comes back set (the sorted case
      -- every PyG batch vector -- pays nothing); the others refuse unsorted input loudly;
    * ``F.one_hot(ligand_v, num_classes)`` (models/molopt_score_model.py:317) and ``index_to_log_onehot`` (:125) raise
      on out-of-range atom types; the kernels would clamp them silently -> same check here.
    One host sync per call (the reference syncs at :316 anyway): the checks reduce to one small device tensor."""
    flags = []
    for b in (batch_protein, batch_ligand):
        flags.append((b[1:] < b[:-1]).any() if b.numel() > 1 else torch.zeros((), dtype=torch.bool, device=b.device))
    if ligand_v is not None and ligand_v.numel():
        flags.append(((ligand_v < 0) | (ligand_v >= num_classes)).any())
    bad = torch.stack([f.to(flags[0].device) for f in flags]).tolist()             # the one synchronisation
    if not allow_unsorted:
        for name, is_bad in zip(('batch_protein', 'batch_ligand'), bad[:2]):
            if is_bad:
                raise ValueError(f'{name} must be sorted by graph id (PyG batch vectors are); got an unsorted vector')
    if len(bad) > 2 and bad[2]:
        raise ValueError(f'ligand_v must be in [0, {num_classes}); got values in '
                         f'[{int(ligand_v.min())}, {int(ligand_v.max())}]')
    return bool(bad[0]), bool(bad[1])


def _stable_order(batch):
    """compose_context's order of one node kind: graphs ascending, the nodes of a graph in their original relative order
    (``torch.sort(..., stable=True)``, models/common.py:126; protein nodes precede ligand nodes inside a graph because the
    concatenation puts them first).  Device-side."""
    return torch.sort(batch, stable=True).indices


def _check_sorted(batch, name='batch'):
    """The refine_net / likelihood / EGNN seams build CSR offsets from ``batch`` as well: refuse an unsorted vector there too."""
    if batch.numel() > 1 and bool((batch[1:] < batch[:-1]).any()):
        raise ValueError(f'{name} must be sorted by graph id (PyG batch vectors are); got an unsorted vector')


# ------------------------------------------------------------------------------------------ parameter holders
class _Offsets(nn.Module):
    """Holds the ``offset`` buffer of GaussianSmearing (models/common.py:7-19, fixed_offset=True)."""

    def __init__(self, num_gaussians: int):
        super().__init__()
        if num_gaussians != len(_FIXED_OFFSETS):
            raise NotImplementedError('the HIP kernels are built for the 20 fixed Gaussian centres')
        self.register_buffer('offset', torch.tensor(_FIXED_OFFSETS, dtype=torch.float32))


class _MLPParams(nn.Module):
    """Parameters of the reference 2-layer MLP (models/common.py:60-80): keys net.0 / net.1 / net.3."""

    def __init__(self, in_dim: int, out_dim: int, hidden_dim: int):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(in_dim, hidden_dim), nn.LayerNorm(hidden_dim), nn.ReLU(),
                                 nn.Linear(hidden_dim, out_dim))

    def forward(self, *a, **k):
        raise RuntimeError('parameter holder: the arithmetic runs in libtargetdiff_hip.so')


class _GateParams(nn.Sequential):
    """``ew_net`` of a stage with ew_net_type 'r': Linear(4 * num_r_gaussian, 1) + Sigmoid (models/uni_transformer.py:34-35, 102-103)."""

    def __init__(self, r_feat_dim):
        super().__init__(nn.Linear(r_feat_dim, 1), nn.Sigmoid())

    def forward(self, *a, **k):
        raise RuntimeError('parameter holder: the arithmetic runs in libtargetdiff_hip.so')


class _X2HParams(nn.Module):
    def __init__(self, hidden, heads, kv_in, r_feat_dim=80, ew_net_type='global', out_fc=False):
        super().__init__()
        self.hk_func = _MLPParams(kv_in, hidden, hidden)
        self.hv_func = _MLPParams(kv_in, hidden, hidden)
        self.hq_func = _MLPParams(hidden, hidden, hidden)
        if ew_net_type == 'r':
            self.ew_net = _GateParams(r_feat_dim)
        elif ew_net_type == 'm':                                     # the gate from the value vector (:36-37)
            self.ew_net = _GateParams(hidden)
        if out_fc:
            self.node_output = _MLPParams(2 * hidden, hidden, hidden)


class _H2XParams(nn.Module):
    def __init__(self, hidden, heads, kv_in, r_feat_dim=80, ew_net_type='global'):
        super().__init__()
        self.xk_func = _MLPParams(kv_in, hidden, hidden)
        self.xv_func = _MLPParams(kv_in, heads, hidden)
        self.xq_func = _MLPParams(hidden, hidden, hidden)
        if ew_net_type == 'r':
            self.ew_net = _GateParams(r_feat_dim)


class _AttLayerParams(nn.Module):
    def __init__(self, hidden, heads, num_r_gaussian, edge_feat_dim, num_x2h, num_h2x, ew_net_type='global', out_fc=False):
        super().__init__()
        kv_in = 2 * hidden + edge_feat_dim + 4 * num_r_gaussian
        self.distance_expansion = _Offsets(num_r_gaussian)
        self.x2h_layers = nn.ModuleList([_X2HParams(hidden, heads, kv_in, 4 * num_r_gaussian, ew_net_type, out_fc) for _ in range(num_x2h)])
        self.h2x_layers = nn.ModuleList([_H2XParams(hidden, heads, kv_in, 4 * num_r_gaussian, ew_net_type) for _ in range(num_h2x)])


class UniTransformerO2TwoUpdateGeneral(nn.Module):
    """``refine_net``: kNN graph + edge gate + num_layers x (x2h, h2x) -- executed by td_refine_forward."""

    def __init__(self, num_blocks, num_layers, hidden_dim, n_heads=1, k=32, num_r_gaussian=50, edge_feat_dim=0,
                 num_node_types=8, act_fn='relu', norm=True, cutoff_mode='radius', ew_net_type='r',
                 num_init_x2h=1, num_init_h2x=0, num_x2h=1, num_h2x=1, r_max=10., x2h_out_fc=True,
                 sync_twoup=False, r=None, max_num_neighbors=32):
        super().__init__()
        unsupported = []
        if not 1 <= int(num_blocks) <= 8: unsupported.append(f'num_blocks={num_blocks} (1..8)')
        if cutoff_mode not in capi.CUTOFF_MODES: unsupported.append(f'cutoff_mode={cutoff_mode!r}')
        if ew_net_type != 'global' and (cutoff_mode not in ('knn', 'radius') or (cutoff_mode == 'knn' and k > 32) or
                                        (cutoff_mode == 'radius' and max_num_neighbors > 32)):
            unsupported.append(f'ew_net_type={ew_net_type!r} on a graph wider than 32 slots per node')
        if act_fn != 'relu' or not norm: unsupported.append(f'act_fn={act_fn!r}/norm={norm}')
        if not (1 <= num_x2h <= 4 and 1 <= num_h2x <= 4): unsupported.append(f'num_x2h={num_x2h}/num_h2x={num_h2x} (1..4 each)')
        if sync_twoup and (num_x2h != 1 or num_h2x != 1): unsupported.append('sync_twoup with several stages per layer')
        if (hidden_dim, n_heads, num_r_gaussian, edge_feat_dim) != (128, 16, 20, 4):
            unsupported.append(f'shape {(hidden_dim, n_heads, num_r_gaussian, edge_feat_dim)}')
        if not 1 <= k <= capi.MAX_FANIN: unsupported.append(f'knn={k} (1..{capi.MAX_FANIN})')
        if unsupported:
            raise NotImplementedError('libtargetdiff_hip.so is built for the live architecture of '
                                      'configs/training.yml:9-42; unsupported: ' + ', '.join(unsupported))
        # graph construction (:276-286) is a run-time choice: knn (any k <= 64; k = 32 is the live configuration and the
        # fast path), hybrid (models/common.py:165-212), radius (dead code in the reference: `self.r` is never assigned,
        # :278 -- here r defaults to r_max and the fan-out is capped at max_num_neighbors, torch_geometric's default 32)
        self.r = float(r if r is not None else r_max)
        self.max_num_neighbors = int(max_num_neighbors)
        if cutoff_mode == 'radius':
            import warnings
            warnings.warn("cutoff_mode='radius' has no counterpart in the reference (it raises there: `self.r` is never "
                          'assigned, models/uni_transformer.py:278).  Here: the first max_num_neighbors same-graph nodes in '
                          f'index order within r = {self.r} A (r defaults to r_max) -- the project\'s own rule, held to '
                          'oracle/shims.py only, not to torch_cluster.radius_graph.', stacklevel=3)
        self.num_blocks, self.num_layers, self.hidden_dim, self.n_heads, self.k = num_blocks, num_layers, hidden_dim, n_heads, k
        self.num_r_gaussian, self.edge_feat_dim = num_r_gaussian, edge_feat_dim
        self.cutoff_mode, self.ew_net_type, self.x2h_out_fc, self.sync_twoup = cutoff_mode, ew_net_type, bool(x2h_out_fc), bool(sync_twoup)
        self.num_x2h, self.num_h2x = int(num_x2h), int(num_h2x)
        self.distance_expansion = _Offsets(num_r_gaussian)
        if ew_net_type == 'global':                                  # :241-242 (the per-stage gates of 'r' live in the layers)
            self.edge_pred_layer = _MLPParams(num_r_gaussian, 1, hidden_dim)
        # built by the reference with (num_init_x2h, num_init_h2x) and never called (uni_transformer.py:245 vs
        # :301-328); kept so that checkpoints load with strict=True.
        self.init_h_emb_layer = _AttLayerParams(hidden_dim, n_heads, num_r_gaussian, edge_feat_dim,
                                                num_init_x2h, num_init_h2x, ew_net_type, bool(x2h_out_fc))
        self.base_block = nn.ModuleList([
            _AttLayerParams(hidden_dim, n_heads, num_r_gaussian, edge_feat_dim, num_x2h, num_h2x, ew_net_type, bool(x2h_out_fc))
            for _ in range(num_layers)])
        self._owner = None       # set by ScorePosNet3D: the module that owns the packed native weights

    def __getstate__(self):
        state = dict(super().__getstate__()) if hasattr(nn.Module, '__getstate__') else self.__dict__.copy()
        state['_owner'] = None           # a weakref: not picklable; the owning ScorePosNet3D re-binds it
        return state

    def forward(self, h, x, mask_ligand, batch, return_all=False, fix_x=False):
        """models/uni_transformer.py:301-328 -> {'x', 'h'[, 'all_x', 'all_h']}; with num_blocks == 1 the lists hold the
        block input and the block output (:304-305, :322-323)."""
        if self._owner is None:
            raise RuntimeError('refine_net must be owned by a ScorePosNet3D (it packs the weights for the HIP library)')
        native = self._owner()._native(h.device)
        _check_sorted(batch)
        B = int(batch.max().item()) + 1 if batch.numel() else 0
        node_ptr = native.graph_ptr(batch.contiguous(), B)
        h_in, x_in = h.contiguous().float(), x.contiguous().float()
        out_h, out_x, _, _ = native.refine_forward(h_in, x_in, mask_ligand, node_ptr, fix_x=fix_x)
        outputs = {'x': out_x, 'h': out_h}
        if return_all:
            if self.num_blocks != 1:        # the states between two blocks stay inside the library
                raise NotImplementedError('return_all with num_blocks > 1: only the final state leaves td_refine_forward')
            outputs.update({'all_x': [x_in, out_x], 'all_h': [h_in, out_h]})
        return outputs


def get_refine_net(refine_net_type, config):
    """models/molopt_score_model.py:13-45."""
    g = lambda k, d=None: _cfg_get(config, k, d)
    if refine_net_type == 'egnn':                                                       # :34-42
        from .egnn import EGNN
        return EGNN(num_layers=g('num_layers'), hidden_dim=g('hidden_dim'), edge_feat_dim=g('edge_feat_dim'),
                    num_r_gaussian=1, k=g('knn'), cutoff_mode=g('cutoff_mode'))
    if refine_net_type != 'uni_o2':
        raise ValueError(refine_net_type)
    g = lambda k, d=None: _cfg_get(config, k, d)
    return UniTransformerO2TwoUpdateGeneral(
        num_blocks=g('num_blocks'), num_layers=g('num_layers'), hidden_dim=g('hidden_dim'), n_heads=g('n_heads'),
        k=g('knn'), edge_feat_dim=g('edge_feat_dim'), num_r_gaussian=g('num_r_gaussian'),
        num_node_types=g('num_node_types'), act_fn=g('act_fn'), norm=g('norm'), cutoff_mode=g('cutoff_mode'),
        ew_net_type=g('ew_net_type'), num_x2h=g('num_x2h'), num_h2x=g('num_h2x'), r_max=g('r_max'),
        x2h_out_fc=g('x2h_out_fc'), sync_twoup=g('sync_twoup'), r=g('r'), max_num_neighbors=g('max_num_neighbors', 32))


# ------------------------------------------------------------------------------------------ schedules
def _sigmoid_betas(beta_start, beta_end, T):
    t = np.linspace(-6, 6, T)
    return 1.0 / (1.0 + np.exp(-t)) * (beta_end - beta_start) + beta_start


def _cosine_alphas(T, s):
    steps = T + 1
    grid = np.linspace(0, steps, steps)
    cum = np.cos(((grid / steps) + s) / (1 + s) * np.pi * 0.5) ** 2
    cum = cum / cum[0]
    return np.sqrt(np.clip(cum[1:] / cum[:-1], a_min=0.001, a_max=1.0))


def _const(x):
    return nn.Parameter(torch.from_numpy(np.asarray(x)).float(), requires_grad=False)


class ShiftedSoftplus(nn.Module):
    """Position 1 of ``v_inference`` (models/common.py:156-162); parameter-free, evaluated inside head_kernel."""

    def forward(self, x):
        raise RuntimeError('evaluated inside libtargetdiff_hip.so')


class _SinusoidalPosEmb(nn.Module):
    """Parameterless first stage of the reference's 'sin' time embedding (models/molopt_score_model.py:182-194); a holder only:
    it keeps the indices of the Sequential -- and with them the state_dict keys time_emb.1.* / time_emb.3.* -- as the reference has them."""

    def __init__(self, dim):
        super().__init__()
        self.dim = dim

    def forward(self, x):
        raise NotImplementedError("time_emb_mode='sin' is constructed for checkpoint compatibility only")


class ScorePosNet3D(nn.Module):
    """Drop-in for models/molopt_score_model.py::ScorePosNet3D on the sampling path.

    Training-only members (get_diffusion_loss, likelihood_estimation, ...) are out of scope.
    """

    def __init__(self, config, protein_atom_feature_dim, ligand_atom_feature_dim):
        super().__init__()
        self.config = config
        g = lambda k, d=None: _cfg_get(config, k, d)
        self.model_mean_type = g('model_mean_type')
        self.loss_v_weight = g('loss_v_weight')
        self.sample_time_method = g('sample_time_method')
        if self.model_mean_type not in ('C0', 'noise'):       # :205; 'noise': the network predicts x_t + eps (:663-666)
            raise ValueError(f'model_mean_type={self.model_mean_type!r}')
        if not g('node_indicator', True):
            raise NotImplementedError('node_indicator=False is not built')

        # ---- variance schedules, float64 on the host exactly as the reference builds them (:221-267)
        T = g('num_diffusion_timesteps')
        sched = g('beta_schedule')
        if sched == 'cosine':
            alphas = _cosine_alphas(T, g('pos_beta_s')) ** 2
            betas = 1.0 - alphas
        elif sched == 'sigmoid':
            betas = _sigmoid_betas(g('beta_start'), g('beta_end'), T)
            alphas = 1.0 - betas
        elif sched == 'linear':
            betas = np.linspace(g('beta_start'), g('beta_end'), T, dtype=np.float64)
            alphas = 1.0 - betas
        else:
            raise NotImplementedError(sched)
        cum = np.cumprod(alphas, axis=0)
        cum_prev = np.append(1.0, cum[:-1])
        self.betas = _const(betas)
        self.num_timesteps = self.betas.size(0)
        self.alphas_cumprod = _const(cum)
        self.alphas_cumprod_prev = _const(cum_prev)
        self.sqrt_alphas_cumprod = _const(np.sqrt(cum))
        self.sqrt_one_minus_alphas_cumprod = _const(np.sqrt(1.0 - cum))
        self.sqrt_recip_alphas_cumprod = _const(np.sqrt(1.0 / cum))
        self.sqrt_recipm1_alphas_cumprod = _const(np.sqrt(1.0 / cum - 1))
        post_var = betas * (1.0 - cum_prev) / (1.0 - cum)
        self.posterior_mean_c0_coef = _const(betas * np.sqrt(cum_prev) / (1.0 - cum))
        self.posterior_mean_ct_coef = _const((1.0 - cum_prev) * np.sqrt(alphas) / (1.0 - cum))
        self.posterior_var = _const(post_var)
        pv32 = self.posterior_var.detach().numpy()        # the reference logs the fp32-rounded tensor (:254)
        self.posterior_logvar = _const(np.log(np.append(pv32[1], pv32[1:])))
        if g('v_beta_schedule') != 'cosine':
            raise NotImplementedError(g('v_beta_schedule'))
        log_a = np.log(_cosine_alphas(self.num_timesteps, g('v_beta_s')))
        log_ca = np.cumsum(log_a)
        one_minus = lambda a: np.log(1 - np.exp(a) + 1e-40)
        self.log_alphas_v = _const(log_a)
        self.log_one_minus_alphas_v = _const(one_minus(log_a))
        self.log_alphas_cumprod_v = _const(log_ca)
        self.log_one_minus_alphas_cumprod_v = _const(one_minus(log_ca))
        self.register_buffer('Lt_history', torch.zeros(self.num_timesteps))
        self.register_buffer('Lt_count', torch.zeros(self.num_timesteps))

        # ---- learnable tensors (held here, packed for the HIP kernels on first use)
        self.hidden_dim = g('hidden_dim')
        self.num_classes = ligand_atom_feature_dim
        self.protein_atom_feature_dim = protein_atom_feature_dim
        emb_dim = self.hidden_dim - 1
        self.protein_atom_emb = nn.Linear(protein_atom_feature_dim, emb_dim)
        self.center_pos_mode = g('center_pos_mode')
        # time embedding (:286-303): extra input columns of ligand_atom_emb, fed with a per-graph feature of the time step
        self.time_emb_dim = int(g('time_emb_dim', 0) or 0)
        self.time_emb_mode = g('time_emb_mode', 'simple')
        if self.time_emb_dim > 0:
            if self.time_emb_mode == 'simple':
                self.ligand_atom_emb = nn.Linear(ligand_atom_feature_dim + 1, emb_dim)
            elif self.time_emb_mode == 'sin':
                # The reference CONSTRUCTS this variant (:292-299) and loads such checkpoints with strict=True; its forward then
                # concatenates the per-GRAPH feature time_emb(time_step) [B, dim] with the per-ATOM one-hot [N_l, C] (:326-327, no
                # [batch_ligand]) and fails unless B == N_l.  Same here: the parameter holders exist (the state_dict round-trips), and
                # forward / sampling refuse (_time_bias) -- there is no behaviour to reproduce.
                self.time_emb = nn.Sequential(_SinusoidalPosEmb(self.time_emb_dim), nn.Linear(self.time_emb_dim, self.time_emb_dim * 4),
                                              nn.GELU(), nn.Linear(self.time_emb_dim * 4, self.time_emb_dim))
                self.ligand_atom_emb = nn.Linear(ligand_atom_feature_dim + self.time_emb_dim, emb_dim)
            else:
                raise NotImplementedError(f'time_emb_mode={self.time_emb_mode!r} (the reference knows simple and sin, :288-301)')
        else:
            self.ligand_atom_emb = nn.Linear(ligand_atom_feature_dim, emb_dim)
        self.refine_net_type = g('model_type')
        if self.refine_net_type != 'uni_o2':
            # the reference builds an EGNN here but cannot run it: forward passes fix_x=, which EGNN.forward does not
            # accept (TypeError at models/molopt_score_model.py:349).  Use targetdiff_amd.egnn.EGNN standalone.
            raise NotImplementedError(f"ScorePosNet3D(model_type={self.refine_net_type!r}): only 'uni_o2' is reachable")
        self.refine_net = get_refine_net(self.refine_net_type, config)
        import weakref
        self.refine_net._owner = weakref.ref(self)
        self.v_inference = nn.Sequential(nn.Linear(self.hidden_dim, self.hidden_dim), ShiftedSoftplus(),
                                         nn.Linear(self.hidden_dim, ligand_atom_feature_dim))
        self._native_model = None
        self._native_key = None
        self._native_options = {}        # td_model_set_option values: re-applied whenever the native handle is rebuilt

    def set_native_option(self, name: str, value: int):
        """Per-model switch of libtargetdiff_hip.so (td_model_set_option: node_proj_split, edge_key_split, h2x_fused,
        session_* ...).  Kept on the module, so it survives every rebuild of the native handle (load_state_dict, .to(),
        an optimizer step, deepcopy / pickle) -- options set directly on ``_native(dev)`` would be lost there."""
        self._native_options[str(name)] = int(value)
        if self._native_model is not None:
            self._native_model.set_option(name, int(value))

    # ------------------------------------------------------------------------------------------ copy / pickle
    def __getstate__(self):
        """The native handle (a ctypes pointer into libtargetdiff_hip.so) is per process and rebuilt on first use:
        drop it so that copy.deepcopy / pickle / torch.save(model) work after a forward has run."""
        state = dict(super().__getstate__()) if hasattr(nn.Module, '__getstate__') else self.__dict__.copy()
        state['_native_model'] = None
        state['_native_key'] = None
        return state

    def __setstate__(self, state):
        super().__setstate__(state)
        import weakref
        self.refine_net._owner = weakref.ref(self)       # re-bind: the copy's refine_net must pack the copy's weights

    def __deepcopy__(self, memo):
        import copy
        new = self.__class__.__new__(self.__class__)
        memo[id(self)] = new
        new.__setstate__(copy.deepcopy(self.__getstate__(), memo))
        return new

    # ------------------------------------------------------------------------------------------ native handle
    def _param_fingerprint(self, device):
        return (str(device),) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def _native(self, device) -> capi.NativeModel:
        """Packed weights inside libtargetdiff_hip.so; rebuilt when a parameter was modified / reloaded."""
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError(f'targetdiff_amd runs on HIP devices only (got {device}); there is no CPU path')
        key = self._param_fingerprint(device)
        if self._native_model is None or key != self._native_key:
            rn = self.refine_net
            cfg = dict(hidden_dim=rn.hidden_dim, n_heads=rn.n_heads, knn=rn.k, num_layers=rn.num_layers,
                       num_r_gaussian=rn.num_r_gaussian, edge_feat_dim=rn.edge_feat_dim,
                       protein_feat_dim=self.protein_atom_feature_dim, ligand_num_classes=self.num_classes,
                       num_timesteps=self.num_timesteps, cutoff_mode=rn.cutoff_mode, radius=rn.r,
                       max_num_neighbors=rn.max_num_neighbors)
            sched = {k: getattr(self, k).detach().cpu().numpy() for k in capi.SCHEDULE_ORDER + capi.SCHEDULE_OPTIONAL}
            cfg['model_mean_type'] = self.model_mean_type
            cfg['num_blocks'] = int(rn.num_blocks)
            cfg['ew_net_type'], cfg['x2h_out_fc'], cfg['sync_twoup'] = rn.ew_net_type, rn.x2h_out_fc, rn.sync_twoup
            cfg['num_x2h'], cfg['num_h2x'] = rn.num_x2h, rn.num_h2x
            sd = self.state_dict()
            if self.time_emb_dim > 0:         # the kernels embed the one-hot part; the time columns go through _time_bias
                sd = dict(sd)
                sd['ligand_atom_emb.weight'] = sd['ligand_atom_emb.weight'][:, :self.num_classes].contiguous()
            self._native_model = capi.NativeModel(cfg, sd, sched, device=device)
            for name, value in getattr(self, '_native_options', {}).items():
                self._native_model.set_option(name, value)
            self._native_key = key
        return self._native_model

    # ------------------------------------------------------------------------------------------ forward
    def forward(self, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, batch_ligand,
                time_step=None, return_all=False, fix_x=False):
        """One denoiser evaluation (models/molopt_score_model.py:313-368).  ``time_step`` [B] is unused by the
        network when time_emb_dim == 0, as in the reference; with a time embedding it is required."""
        if return_all and self.refine_net.num_blocks != 1:
            raise NotImplementedError('return_all with num_blocks > 1: only the final state leaves the library')
        native = self._native(protein_pos.device)
        B = int(batch_protein.max().item()) + 1          # same host sync as the reference (:316)
        gbias = self._time_bias(time_step, B)
        unsorted_p, unsorted_l = _check_graph_inputs(batch_protein, batch_ligand, init_ligand_v, self.num_classes, allow_unsorted=True)
        # compose_context (models/common.py:120-137) accepts unsorted batch vectors: it stable-sorts the nodes by graph and the
        # outputs come back in THAT order (`final_pos[mask_ligand]`, :352 -- the ligand rows are not put back in input order)
        if unsorted_p:
            o = _stable_order(batch_protein)
            protein_pos, protein_v, batch_protein = protein_pos[o], protein_v[o], batch_protein[o]
        if unsorted_l:
            o = _stable_order(batch_ligand)
            init_ligand_pos, init_ligand_v, batch_ligand = init_ligand_pos[o], init_ligand_v[o], batch_ligand[o]
        pptr = native.graph_ptr(batch_protein.contiguous(), B)
        lptr = native.graph_ptr(batch_ligand.contiguous(), B)
        lpos, lv = init_ligand_pos.contiguous().float(), init_ligand_v.contiguous()
        preds = native.model_forward(protein_pos.contiguous().float(), protein_v.contiguous().float(), pptr, lpos, lv, lptr,
                                     fix_x=fix_x, ligand_graph_bias=gbias)
        if return_all:
            # :360-367 -- the refine net records the state before and after each block; num_blocks == 1 here, so the
            # lists hold the block input (the embedded ligand atoms at their input positions) and the block output
            preds['layer_pred_ligand_pos'] = [lpos.clone(), preds['pred_ligand_pos']]
            emb0 = native.embed_ligand(lv)
            if gbias is not None:
                emb0 = emb0 + gbias[batch_ligand]
            preds['layer_pred_ligand_v'] = [native.v_inference(emb0), preds['pred_ligand_v']]
        return preds

    def _time_bias(self, time_step, B):
        """The time-embedding term of the ligand atoms' embedding, one row per graph ([B, 128] fp32, the node-indicator column
        zero), or None when time_emb_dim == 0 (models/molopt_score_model.py:319-329): ligand_atom_emb is linear in its input
        ``[one_hot(v), time feature]``, so its time columns applied to the graph's time feature are a per-graph vector added to
        every ligand atom's embedding -- a few hundred multiplies, done here with torch; the kernels add the row."""
        if self.time_emb_dim > 0 and self.time_emb_mode == 'sin':
            raise NotImplementedError("time_emb_mode='sin': the reference's forward concatenates a per-graph with a per-atom tensor "
                                      '(models/molopt_score_model.py:326-327) and fails unless every graph has one ligand atom; not reproduced')
        if self.time_emb_dim <= 0:
            return None
        if time_step is None:
            raise ValueError('time_emb_dim > 0: forward needs time_step (one entry per graph)')
        W = self.ligand_atom_emb.weight                                       # [127, C + 1]
        t = time_step.to(W.device)
        if t.numel() != B:
            raise ValueError(f'time_step has {t.numel()} entries for {B} graphs')
        feat = (t / self.num_timesteps).to(torch.float32).unsqueeze(-1)           # :321-324 ('simple', the only mode that runs)
        out = torch.zeros(B, self.hidden_dim, dtype=torch.float32, device=W.device)
        out[:, :W.shape[0]] = feat.to(torch.float32) @ W[:, self.num_classes:].t().to(torch.float32)
        return out

    @torch.no_grad()
    def likelihood_estimation(self, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, batch_ligand, time_step,
                              noise_source=None):
        """models/molopt_score_model.py:565-613 (scripts/likelihood_est_diffusion.py:30,48): per-graph ``(kl_pos, kl_v)``
        at ``time_step`` [B], or the prior terms when every entry equals ``num_timesteps``.  ``noise_source(0, name,
        like)`` may inject the Gaussian / uniform draws (parity tests); default: torch RNG in the reference's order."""
        native = self._native(protein_pos.device)
        T = self.num_timesteps
        _check_graph_inputs(batch_protein, batch_ligand, ligand_v, self.num_classes)
        B = int(batch_protein.max().item()) + 1
        pptr = native.graph_ptr(batch_protein.contiguous(), B)
        lptr = native.graph_ptr(batch_ligand.contiguous(), B)
        ppos = protein_pos.detach().clone().contiguous().float()
        lpos = ligand_pos.detach().clone().contiguous().float()
        lv = ligand_v.contiguous()
        native.center_pos(ppos, pptr, lpos, lptr)                                          # :568 (mode='protein')
        all_T, all_lt = bool((time_step == T).all()), bool((time_step < T).all())
        assert all_T or all_lt                                                             # :570
        if all_T:
            assert int(batch_ligand.max().item()) < self.num_classes                       # index_to_log_onehot's check
            return native.likelihood_prior(lptr, lpos, batch_ligand.contiguous())          # :571-575 (sic: batch_ligand)
        t32 = time_step.to(torch.int32).contiguous()
        if noise_source is None:
            noise = torch.zeros_like(lpos).normal_()                                       # :579-580
            uniform = torch.rand(lpos.shape[0], self.num_classes, dtype=torch.float32, device=lpos.device)   # :161
        else:
            noise = noise_source(0, 'noise', lpos)
            uniform = noise_source(0, 'uniform', None)
        pos_t, v_t = native.perturb(t32, lptr, lpos, lv, noise, uniform)                    # :582-586
        if self.model_mean_type != 'C0':
            raise ValueError(self.model_mean_type)                                # :601-605: the reference raises here as well
        preds = native.model_forward(ppos, protein_v.contiguous().float(), pptr, pos_t, v_t, lptr, want_final_h=False,
                                     ligand_graph_bias=self._time_bias(time_step, B))
        return native.likelihood_terms(t32, lptr, lpos, pos_t, lv, v_t, preds['pred_ligand_pos'], preds['pred_ligand_v'])

    @torch.no_grad()
    def fetch_embedding(self, protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, batch_ligand):
        """models/molopt_score_model.py:620-631."""
        return self(protein_pos, protein_v, batch_protein, ligand_pos, ligand_v, batch_ligand, fix_x=True)

    # ------------------------------------------------------------------------------------------ sampling
    def begin_sampling(self, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, batch_ligand,
                       num_steps=None, center_pos_mode=None, max_graph_nodes=0, noise_source=None, use_session=True,
                       pos_only=False, generator=None, use_graph=None):
        """Set up the reverse-diffusion state on the device and return a :class:`ReverseSampler`
        (``.step()`` = one iteration of the loop at models/molopt_score_model.py:650-693)."""
        return ReverseSampler(self, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v,
                              batch_ligand, num_steps, center_pos_mode, max_graph_nodes, noise_source, use_session,
                              pos_only, generator, use_graph)

    @torch.no_grad()
    def sample_diffusion(self, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, batch_ligand,
                         num_steps=None, center_pos_mode=None, pos_only=False, max_graph_nodes=0,
                         noise_source=None, use_session=True, use_graph=None):
        """Ancestral sampling loop (models/molopt_score_model.py:633-703).

        Differences from the reference are confined to *where* things run, not what is computed: no
        ``.item()`` / D2H sync inside the loop (graph offsets are built once), trajectories are
        accumulated in device buffers and copied to the host once at the end (same returned lists of CPU
        tensors, positions de-centred).  ``noise_source(step, name, like)`` may inject the Gaussian /
        uniform draws (parity tests); by default torch.randn_like / rand_like are used in the reference's
        order.  ``use_session=False`` evaluates the stateless td_model_forward at every step (no static-protein
        caching); the two are bit-identical (tests/test_gpu_long_parity.py).  With a session the ~50 launches of a step are
        one replayable unit (td_session_step): ``use_graph=True`` captures them once into a hipGraph and replays it, False issues
        them one by one, None (default) replays when the caller runs on a real stream -- the same kernels with the same arguments
        either way, hence the same bits (tests/test_gpu_step_graph.py)."""
        sampler = self.begin_sampling(protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v,
                                      batch_ligand, num_steps, center_pos_mode, max_graph_nodes, noise_source,
                                      use_session=use_session, pos_only=pos_only, use_graph=use_graph)
        while not sampler.done:
            sampler.step()
        return sampler.finish()


class ReverseSampler:
    """Device-resident state of one ``sample_diffusion`` call."""

    @torch.no_grad()
    def __init__(self, model, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, batch_ligand,
                 num_steps, center_pos_mode, max_graph_nodes, noise_source, use_session=True, pos_only=False, generator=None,
                 use_graph=None):
        self.pos_only = bool(pos_only)
        self.generator = generator          # None: torch's global generator (the reference's stream of draws)
        if center_pos_mode not in ('protein', 'none'):
            # center_pos (models/molopt_score_model.py:110-120) raises for anything else -- including the signature
            # default None, which would otherwise sample un-centred (off-distribution) without a word
            raise NotImplementedError(f'center_pos_mode={center_pos_mode!r}: pass \'protein\' (configs/sampling.yml) or \'none\'')
        unsorted_p, unsorted_l = _check_graph_inputs(batch_protein, batch_ligand, init_ligand_v, model.num_classes, allow_unsorted=True)
        dev = protein_pos.device
        self.native = native = model._native(dev)
        T = model.num_timesteps
        num_steps = T if num_steps is None else num_steps
        self.B = B = int(batch_protein.max().item()) + 1                                 # :638 (once, not per step)
        if unsorted_p:              # the protein never changes: put it in compose_context's order once
            o = _stable_order(batch_protein)
            protein_pos, protein_v, batch_protein = protein_pos[o], protein_v[o], batch_protein[o]
        # Unsorted ligand vector: the reference's loop keeps its state (ligand_pos / ligand_v, :644-646) in INPUT order while every
        # forward returns its predictions in compose_context's order, and combines the two element by element (:663-685).  Reproduced
        # as it is: the state stays in input order, each step gathers it into graph order for the denoiser (the slow two-call form
        # of the step; a sorted vector -- every PyG batch -- takes the fused step).
        self._lig_order = _stable_order(batch_ligand) if unsorted_l else None
        self.pptr = native.graph_ptr(batch_protein.contiguous(), B)
        self.lptr = native.graph_ptr((batch_ligand[self._lig_order] if unsorted_l else batch_ligand).contiguous(), B)
        self.ppos = protein_pos.detach().clone().contiguous().float()
        self.lpos = init_ligand_pos.detach().clone().contiguous().float()
        self.lv = init_ligand_v.detach().clone().contiguous()
        self.pv = protein_v.contiguous().float()
        self.batch_ligand = batch_ligand
        self.Nl, self.C = self.lpos.shape[0], model.num_classes
        self.offset = None
        if center_pos_mode == 'protein':
            if unsorted_l:          # :113-118 with the ligand in input order: the per-graph protein centroid, taken off atom by atom
                self.offset = native.center_pos(self.ppos, self.pptr, None, self.lptr)
                self.lpos -= self.offset[batch_ligand]
            else:
                self.offset = native.center_pos(self.ppos, self.pptr, self.lpos, self.lptr)   # :642
        steps = list(reversed(range(T - num_steps, T)))                                   # :649
        self.S = S = len(steps)
        Nl, C = self.Nl, self.C
        self.pos_traj = torch.empty(S, Nl, 3, dtype=torch.float32, device=dev)
        self.v_traj = torch.empty(S, Nl, dtype=torch.int64, device=dev)
        # pos_only (:681): the types are frozen, no uniforms are drawn and v0 / vt are not recorded
        SV = 0 if self.pos_only else S
        self.v0_traj = torch.empty(SV, Nl, C, dtype=torch.float32, device=dev)
        self.vt_traj = torch.empty(SV, Nl, C, dtype=torch.float32, device=dev)
        self._half = torch.full((Nl, C), 0.5, dtype=torch.float32, device=dev) if self.pos_only else None
        self._v_scratch = torch.empty(Nl, dtype=torch.int64, device=dev) if self.pos_only else None
        self.t_all = torch.tensor(steps, dtype=torch.int32, device=dev).view(S, 1).expand(S, B).contiguous()
        self.max_graph_nodes = max_graph_nodes
        self.noise_source = noise_source
        self.bufs = {}
        self.s = 0
        # this step's draws live in fixed buffers, refilled in place every step (the captured step reads them by address)
        self.model = model
        self._gbias = None
        if model.time_emb_dim > 0:           # this step's time-embedding rows, in a fixed buffer (the captured step reads it by address)
            self._gbias = torch.zeros(B, model.hidden_dim, dtype=torch.float32, device=dev)
        self._noise = torch.empty(Nl, 3, dtype=torch.float32, device=dev)
        self._uniform = self._half if self.pos_only else torch.empty(Nl, C, dtype=torch.float32, device=dev)
        # loop-invariant protein state lives in a native session (td_session); opt out with use_session=False
        self.session = None
        self._io = None
        self.use_graph = use_graph
        if use_session and self.Nl > 0 and self.ppos.shape[0] > 0:
            self.session = capi.NativeSession(native, self.ppos, self.pv, self.pptr, self.lptr, self.Nl, max_graph_nodes)
            # the whole step (forward + posterior + trajectory record) is one replayable unit: its per-step arguments (step
            # index -> time step and trajectory slot, current state, draws) sit in device memory
            self._step_index = torch.zeros(2, dtype=torch.int32, device=dev)
            if S > 0:
                self._io = self.session.make_step_io(self._step_index, self.t_all, self.lpos, self.lv, self._noise,
                                                     self._uniform, self.pos_traj, self.v_traj, self.v0_traj, self.vt_traj,
                                                     self.pos_only, ligand_graph_bias=self._gbias)

    def _graph_now(self):
        """Replay the step as a captured hipGraph?  ``use_graph=None`` (default): when the caller runs on a real stream (the
        overlapped batches of sample_diffusion_ligand, a serving loop with its own stream) -- a step cannot be captured on the
        device's legacy default stream, and moving it to a side stream behind event fences costs more (C1: +10 %) than the
        replay saves there."""
        if self.use_graph is not None:
            return bool(self.use_graph)
        dev = self.lpos.device
        return dev.type == 'cuda' and torch.cuda.current_stream(dev) != torch.cuda.default_stream(dev)

    @property
    def done(self):
        return self.s >= self.S

    def _draw(self, s):
        """This step's Gaussian / uniform draws into the fixed buffers, in the reference's order (:677, then :161); with a time
        embedding, also this step's per-graph embedding rows (:652 time_step = t for every graph)."""
        if self._gbias is not None:
            self._gbias.copy_(self.model._time_bias(self.t_all[s], self.B))
        if self.noise_source is None:
            # == torch.randn_like(lpos) / torch.rand(Nl, C): the same generator stream, written in place
            self._noise.normal_(generator=self.generator)
            if not self.pos_only:
                self._uniform.uniform_(generator=self.generator)
        else:
            self._noise.copy_(self.noise_source(s, 'noise', self.lpos))
            if not self.pos_only:
                self._uniform.copy_(self.noise_source(s, 'uniform', self.v0_traj[s]))

    @torch.no_grad()
    def step(self):
        s, native = self.s, self.native
        if self.session is not None and self.S > 0 and self._lig_order is None:
            self._draw(s)
            self.session.step(self._io, use_graph=self._graph_now())     # lpos / lv are updated in place, slot s of the trajectories filled
            self.s += 1
            return
        lpos_in, lv_in = self.lpos, self.lv
        if self._lig_order is not None:
            lpos_in, lv_in = self.lpos[self._lig_order].contiguous(), self.lv[self._lig_order].contiguous()
        self._draw(s)
        if self.session is not None:
            preds = self.session.forward(lpos_in, lv_in, out=self.bufs, ligand_graph_bias=self._gbias)
        else:
            preds = native.model_forward(self.ppos, self.pv, self.pptr, lpos_in, lv_in, self.lptr,
                                         max_graph_nodes=self.max_graph_nodes, want_final_h=False, out=self.bufs,
                                         ligand_graph_bias=self._gbias)
        self.bufs = preds
        if self.pos_only:
            native.posterior_step(self.t_all[s], self.lptr, self.lpos, self.lv, preds['pred_ligand_pos'],
                                  preds['pred_ligand_v'], self._noise, self._half, pos_next=self.pos_traj[s],
                                  v_next=self._v_scratch)
            self.v_traj[s].copy_(self.lv)                                                  # :689
            self.lpos = self.pos_traj[s]
            self.s += 1
            return
        native.posterior_step(self.t_all[s], self.lptr, self.lpos, self.lv, preds['pred_ligand_pos'],
                              preds['pred_ligand_v'], self._noise, self._uniform, pos_next=self.pos_traj[s],
                              v_next=self.v_traj[s], log_v0=self.v0_traj[s], log_post=self.vt_traj[s])
        self.lpos, self.lv = self.pos_traj[s], self.v_traj[s]
        self.s += 1

    @torch.no_grad()
    def finish(self):
        S = self.s
        SV = 0 if self.pos_only else S
        pos_traj, v_traj, v0_traj, vt_traj = self.pos_traj[:S], self.v_traj[:S], self.v0_traj[:SV], self.vt_traj[:SV]
        shift = self.offset[self.batch_ligand] if self.offset is not None else None
        if shift is not None:
            pos_traj = pos_traj + shift.unsqueeze(0)                                       # :691
        final_pos = pos_traj[-1].clone() if S else (self.lpos + shift if shift is not None else self.lpos.clone())
        final_v = v_traj[-1].clone() if S else self.lv
        pos_cpu, v_cpu, v0_cpu, vt_cpu = pos_traj.cpu(), v_traj.cpu(), v0_traj.cpu(), vt_traj.cpu()
        return {'pos': final_pos, 'v': final_v, 'pos_traj': list(pos_cpu.unbind(0)), 'v_traj': list(v_cpu.unbind(0)),
                'v0_traj': list(v0_cpu.unbind(0)), 'vt_traj': list(vt_cpu.unbind(0))}
