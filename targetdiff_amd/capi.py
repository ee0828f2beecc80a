"""ctypes binding of libtargetdiff_hip.so (C ABI: include/targetdiff_hip.h).

This is the binding a maintainer of the reference would add: torch tensors own all
device memory, the library only sees raw device pointers + sizes + the HIP stream of the current torch
stream.  There is no CPU fallback anywhere in this package: if the library is missing or a tensor is
not on a HIP device the call raises.
"""
from __future__ import annotations

import ctypes
import os
from ctypes import POINTER, c_char_p, c_float, c_int32, c_int64, c_size_t, c_void_p

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libtargetdiff_hip.so')

TD_OK = 0
HIDDEN = 128
KNN = 32


class TdConfig(ctypes.Structure):
    _fields_ = [('hidden_dim', c_int32), ('n_heads', c_int32), ('knn', c_int32), ('num_layers', c_int32),
                ('num_r_gaussian', c_int32), ('edge_feat_dim', c_int32), ('protein_feat_dim', c_int32),
                ('ligand_num_classes', c_int32), ('num_timesteps', c_int32), ('cutoff_mode', c_int32), ('radius', c_float),
                ('max_num_neighbors', c_int32), ('model_mean_type', c_int32), ('num_blocks', c_int32), ('ew_net_type', c_int32),
                ('x2h_out_fc', c_int32), ('sync_twoup', c_int32), ('num_x2h', c_int32), ('num_h2x', c_int32), ('reserved', c_int32 * 1)]


CUTOFF_MODES = {'knn': 0, 'hybrid': 1, 'radius': 2}      # TD_CUTOFF_* (include/targetdiff_hip.h)
MAX_FANIN = 64
ABI_VERSION = 5


# every symbol include/targetdiff_hip.h declares: (restype, argtypes)
_P = c_void_p
SIGNATURES = {
    'td_abi_version': (c_int32, []),
    'td_last_error': (c_char_p, []),
    'td_model_create': (c_int32, [POINTER(TdConfig), POINTER(c_float), c_size_t, POINTER(c_float), c_size_t,
                                  POINTER(c_void_p)]),
    'td_model_destroy': (None, [_P]),
    'td_model_num_weights': (c_size_t, [POINTER(TdConfig)]),
    'td_model_set_option': (c_int32, [_P, c_char_p, c_int32]),
    'td_model_get_option': (c_int32, [_P, c_char_p, POINTER(c_int32)]),
    'td_graph_build': (c_int32, [_P, _P, _P, _P, c_int64, c_int64, c_int32, _P, c_int32, _P]),
    'td_workspace_bytes': (c_size_t, [_P, c_int64, c_int64, c_int64]),
    'td_graph_ptr': (c_int32, [_P, c_int64, c_int64, _P, _P]),
    'td_knn': (c_int32, [_P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P]),
    'td_refine_forward': (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P, _P, _P, _P,
                                    c_size_t, _P]),
    'td_model_forward': (c_int32, [_P, _P, _P, _P, c_int64, _P, _P, _P, c_int64, c_int64, c_int32, c_int32, _P, _P,
                                   _P, _P, _P, c_size_t, _P, _P]),
    'td_posterior_step': (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'td_center_pos': (c_int32, [_P, _P, _P, _P, c_int64, _P, c_int32, c_int32, _P]),
    'td_perturb': (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P]),
    'td_likelihood_terms': (c_int32, [_P, _P, _P, c_int64, c_int64, _P, _P, _P, _P, _P, _P, _P, _P, _P]),
    'td_likelihood_prior': (c_int32, [_P, _P, c_int64, c_int64, _P, _P, _P, _P, _P]),
    'td_embed_ligand': (c_int32, [_P, _P, c_int64, _P, _P]),
    'td_v_inference': (c_int32, [_P, _P, c_int64, _P, _P]),
    'td_egnn_num_weights': (ctypes.c_size_t, [c_int32]),
    'td_egnn_create': (c_int32, [c_int32, c_int32, c_int32, c_int32, POINTER(c_float), ctypes.c_size_t, POINTER(c_void_p)]),
    'td_egnn_destroy': (None, [_P]),
    'td_egnn_workspace_bytes': (ctypes.c_size_t, [c_int64]),
    'td_egnn_forward': (c_int32, [_P, _P, _P, _P, _P, c_int64, c_int64, c_int32, _P, _P, _P, _P, _P, ctypes.c_size_t, _P]),
    'td_session_create': (c_int32, [_P, _P, _P, _P, c_int64, _P, c_int64, c_int64, c_int32, _P, POINTER(c_void_p)]),
    'td_session_destroy': (None, [_P]),
    'td_session_forward': (c_int32, [_P, _P, _P, _P, _P, _P, _P, _P]),
    'td_session_row_counts': (c_int32, [_P, POINTER(c_int32), c_int32, _P]),
    'td_session_step': (c_int32, [_P, _P, c_int32, _P]),
    'td_session_step_graph': (c_int32, [_P]),
    'td_build_tag': (ctypes.c_char_p, []),
    'td_debug_fail_alloc': (c_int32, [c_int32]),
    'td_debug_node_stage': (c_int32, [_P, c_int32, c_int32, _P, c_int64, _P, _P, _P]),
    'td_debug_reductions': (c_int32, [_P, _P, _P]),
    'td_debug_wg_trace': (c_int32, [_P, c_int32]),
    'td_profile_begin': (c_int32, [ctypes.c_uint32]),
    'td_profile_end': (c_int32, [POINTER(c_float), POINTER(c_int32), c_int32]),
}

class StepIO(ctypes.Structure):
    """td_step_io of include/targetdiff_hip.h: the per-step arguments of td_session_step, all in device memory"""
    _fields_ = [('d_step', c_void_p), ('d_t_all', c_void_p), ('num_steps', c_int32), ('pos_only', c_int32),
                ('d_ligand_pos', c_void_p), ('d_ligand_v', c_void_p), ('d_noise', c_void_p), ('d_uniform', c_void_p),
                ('d_pos_traj', c_void_p), ('d_v_traj', c_void_p), ('d_v0_traj', c_void_p), ('d_vt_traj', c_void_p),
                ('d_ligand_graph_bias', c_void_p)]


PROFILE_CLASSES = ('knn', 'gate', 'node_proj', 'x2h_k', 'x2h_v', 'h2x_k', 'h2x_v', 'compose', 'head', 'posterior')


def profile_begin(classes=PROFILE_CLASSES):
    mask = 0
    for c in classes:
        mask |= 1 << PROFILE_CLASSES.index(c)
    _check(load_library().td_profile_begin(mask), 'td_profile_begin')


def profile_end() -> dict:
    n = len(PROFILE_CLASSES)
    ms = (c_float * n)()
    cnt = (c_int32 * n)()
    _check(load_library().td_profile_end(ms, cnt, n), 'td_profile_end')
    return {name: {'ms': float(ms[i]), 'launches': int(cnt[i])} for i, name in enumerate(PROFILE_CLASSES)}


_lib = None


def build_tag() -> str:
    """td_build_tag(): names the sources the loaded library was built from; ties a measurement (bench line, PMC profile) to them."""
    return load_library().td_build_tag().decode()


def load_library(path: str = LIB_PATH):
    """dlopen the HIP library.  Raises (never falls back) when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(path):
        raise RuntimeError(f'{path} not found: build it with `python -m targetdiff_amd.build` '
                           '(hipcc --offload-arch=gfx950).  There is no CPU fallback.')
    lib = ctypes.CDLL(path)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)          # AttributeError if the symbol is missing
        fn.restype = res
        fn.argtypes = args
    if lib.td_abi_version() != ABI_VERSION:
        raise RuntimeError(f'{path} has ABI version {lib.td_abi_version()}, this binding needs {ABI_VERSION}: rebuild it '
                           '(python -m targetdiff_amd.build)')
    _lib = lib
    return lib


def _check(rc: int, what: str):
    if rc != TD_OK:
        msg = load_library().td_last_error()
        raise RuntimeError(f'{what} failed ({rc}): {msg.decode() if msg else "?"}')


def _ptr(t: torch.Tensor | None, dtype=None, what='tensor'):
    if t is None:
        return None
    if not t.is_cuda:
        raise RuntimeError(f'{what} must live on a HIP device (got {t.device}); targetdiff_amd has no CPU path')
    if dtype is not None and t.dtype != dtype:
        raise TypeError(f'{what} must be {dtype}, got {t.dtype}')
    if not t.is_contiguous():
        raise ValueError(f'{what} must be contiguous')
    return c_void_p(t.data_ptr())


def _graph_bias_ptr(t, B):
    """[B][128] fp32 per-graph term of the ligand embedding (time embedding), or None"""
    if t is None:
        return None
    if t.dim() != 2 or t.shape[1] != HIDDEN or (B is not None and t.shape[0] != B):
        raise ValueError(f'ligand_graph_bias must be [B, {HIDDEN}] (got {tuple(t.shape)})')
    return _ptr(t, torch.float32, 'ligand_graph_bias')


def _stream(device=None):
    """HIP stream of torch's current stream ON `device` (the tensors' device, not the process-wide current device)."""
    return c_void_p(torch.cuda.current_stream(device).cuda_stream)


def _canonical_device(device=None) -> torch.device:
    d = torch.device('cuda') if device is None else torch.device(device)
    if d.type != 'cuda':
        raise RuntimeError(f'targetdiff_amd runs on HIP devices only (got {d}); there is no CPU path')
    return torch.device('cuda', torch.cuda.current_device()) if d.index is None else d


def _on(device):
    """Every library call runs with the tensors' device current: the library launches on / allocates from the current HIP
    device, and the reference CLI allows --device cuda:1 while the current device stays 0."""
    return torch.cuda.device(device)


# ----------------------------------------------------------------------------------------- weight blob
MLP_KEYS = ('net.0.weight', 'net.0.bias', 'net.1.weight', 'net.1.bias', 'net.3.weight', 'net.3.bias')


EW_NET_TYPES = {'global': 0, 'r': 1, 'm': 3}              # td_config.ew_net_type; anything else = 2 (e_w = 1)


def ew_net_code(ew_net_type) -> int:
    return EW_NET_TYPES.get(ew_net_type, 2)


def flat_key_order(num_layers: int, ew_net_type='global', x2h_out_fc=False, num_x2h=1, num_h2x=1):
    """Order of the reference state_dict tensors inside the flat blob td_model_create consumes
    (key names: SURVEY.md Appendix C; init_h_emb_layer and the schedule constants are not part of it).  Per layer: offsets; per x2h stage hk, hv, hq,
    [node_output (x2h_out_fc)], [x2h ew_net ('r': 80 + 1, 'm': 128 + 1)], xk, xv, xq, [h2x ew_net ('r')]; the global gate MLP only with ew_net_type 'global'."""
    keys = ['protein_atom_emb.weight', 'protein_atom_emb.bias', 'ligand_atom_emb.weight', 'ligand_atom_emb.bias',
            'refine_net.distance_expansion.offset']
    if ew_net_type == 'global':
        keys += [f'refine_net.edge_pred_layer.{k}' for k in MLP_KEYS]
    for l in range(num_layers):
        p = f'refine_net.base_block.{l}'
        keys.append(f'{p}.distance_expansion.offset')
        for i in range(num_x2h):
            for f in ('hk_func', 'hv_func', 'hq_func'):
                keys += [f'{p}.x2h_layers.{i}.{f}.{k}' for k in MLP_KEYS]
            if x2h_out_fc:
                keys += [f'{p}.x2h_layers.{i}.node_output.{k}' for k in MLP_KEYS]
            if ew_net_type in ('r', 'm'):
                keys += [f'{p}.x2h_layers.{i}.ew_net.0.weight', f'{p}.x2h_layers.{i}.ew_net.0.bias']
        for j in range(num_h2x):
            for f in ('xk_func', 'xv_func', 'xq_func'):
                keys += [f'{p}.h2x_layers.{j}.{f}.{k}' for k in MLP_KEYS]
            if ew_net_type == 'r':
                keys += [f'{p}.h2x_layers.{j}.ew_net.0.weight', f'{p}.h2x_layers.{j}.ew_net.0.bias']
    keys += ['v_inference.0.weight', 'v_inference.0.bias', 'v_inference.2.weight', 'v_inference.2.bias']
    return keys


def flatten_state_dict(sd, num_layers: int, ew_net_type='global', x2h_out_fc=False, num_x2h=1, num_h2x=1) -> np.ndarray:
    parts = [sd[k].detach().to('cpu', torch.float32).contiguous().reshape(-1).numpy()
             for k in flat_key_order(num_layers, ew_net_type, x2h_out_fc, num_x2h, num_h2x)]
    return np.ascontiguousarray(np.concatenate(parts), dtype=np.float32)


SCHEDULE_ORDER = ('posterior_mean_c0_coef', 'posterior_mean_ct_coef', 'posterior_logvar', 'log_alphas_v',
                  'log_one_minus_alphas_v', 'log_alphas_cumprod_v', 'log_one_minus_alphas_cumprod_v')
# optional 8th array: needed by td_perturb / td_likelihood_prior (likelihood estimation), not by sampling
SCHEDULE_OPTIONAL = ('alphas_cumprod', 'sqrt_recip_alphas_cumprod', 'sqrt_recipm1_alphas_cumprod')
MEAN_TYPES = {'C0': 0, 'noise': 1}                        # td_config.model_mean_type


def _device_bound(fn):
    """Run a method of a handle-owning class with the handle's device current."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **k):
        for t in list(a) + list(k.values()):
            if torch.is_tensor(t) and t.is_cuda and t.device != self.device:
                raise RuntimeError(f'{fn.__name__}: tensor on {t.device}, but the native handle lives on {self.device}')
        with _on(self.device):
            return fn(self, *a, **k)
    return wrapped


class NativeModel:
    """Owns a td_model handle (packed weights on `device`) and a growable workspace."""

    def __init__(self, cfg: dict, state_dict, schedules: dict | None = None, device=None):
        self.lib = load_library()
        if not torch.cuda.is_available():
            raise RuntimeError('no HIP device visible: targetdiff_amd needs an MI355X (gfx950); there is no CPU path')
        self.device = _canonical_device(device)
        mode = cfg.get('cutoff_mode', 'knn')
        if mode not in CUTOFF_MODES:
            raise ValueError(f'cutoff_mode must be one of {sorted(CUTOFF_MODES)}, got {mode!r}')
        self.cfg = TdConfig(hidden_dim=cfg['hidden_dim'], n_heads=cfg['n_heads'], knn=cfg['knn'],
                            num_layers=cfg['num_layers'], num_r_gaussian=cfg['num_r_gaussian'],
                            edge_feat_dim=cfg['edge_feat_dim'], protein_feat_dim=cfg['protein_feat_dim'],
                            ligand_num_classes=cfg['ligand_num_classes'], num_timesteps=cfg['num_timesteps'],
                            cutoff_mode=CUTOFF_MODES[mode], radius=float(cfg.get('radius', 0.0)),
                            max_num_neighbors=int(cfg.get('max_num_neighbors', 32)),
                            model_mean_type=MEAN_TYPES[cfg.get('model_mean_type', 'C0')], num_blocks=int(cfg.get('num_blocks', 1) or 1),
                            ew_net_type=ew_net_code(cfg.get('ew_net_type', 'global')), x2h_out_fc=int(bool(cfg.get('x2h_out_fc', False))),
                            sync_twoup=int(bool(cfg.get('sync_twoup', False))), num_x2h=int(cfg.get('num_x2h', 1) or 1),
                            num_h2x=int(cfg.get('num_h2x', 1) or 1))
        self.cutoff_mode, self.k = mode, int(cfg['knn'])
        self.default_graph = mode == 'knn' and self.k <= KNN       # the 32-slot fast path (and the caching session); k < 32
                                                                   # is the 32-NN row with the slots >= k masked
        self.num_classes = int(cfg['ligand_num_classes'])
        blob = flatten_state_dict(state_dict, cfg['num_layers'], cfg.get('ew_net_type', 'global'), bool(cfg.get('x2h_out_fc', False)),
                                  int(cfg.get('num_x2h', 1) or 1), int(cfg.get('num_h2x', 1) or 1))
        expect = self.lib.td_model_num_weights(ctypes.byref(self.cfg))
        if blob.size != expect:
            raise ValueError(f'weight blob has {blob.size} floats, library expects {expect}')
        sched_ptr, sched_n = None, 0
        if schedules is not None:
            # 7 arrays, + alphas_cumprod (8), + the two 'noise' arrays (10): the library takes these three lengths
            opt = SCHEDULE_OPTIONAL if all(o in schedules for o in SCHEDULE_OPTIONAL) else tuple(o for o in SCHEDULE_OPTIONAL[:1] if o in schedules)
            sch = np.ascontiguousarray(np.concatenate(
                [np.asarray(schedules[k], dtype=np.float32).reshape(-1) for k in SCHEDULE_ORDER + opt]))
            sched_ptr, sched_n = sch.ctypes.data_as(POINTER(c_float)), sch.size
        handle = c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.td_model_create(ctypes.byref(self.cfg), blob.ctypes.data_as(POINTER(c_float)), blob.size,
                                            sched_ptr, sched_n, ctypes.byref(handle)), 'td_model_create')
        self.handle = handle
        self._ws = None

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h and getattr(self, 'lib', None) is not None:
            self.lib.td_model_destroy(h)

    # ------------------------------------------------------------------------------------------
    @_device_bound
    def set_option(self, name: str, value: int):
        """Per-model switch (td_model_set_option): 'node_proj_split', 'edge_key_split', 'edge_first_layer_f16', 'edge_second_layer_f16', 'h2x_fused', 'session_share_pockets', 'session_hop_levels',
        'session_forward_reach', 'session_step_lists'.  Stored in the native handle; nothing is read from the environment."""
        _check(self.lib.td_model_set_option(self.handle, name.encode(), int(value)), 'td_model_set_option')

    def get_option(self, name: str) -> int:
        v = c_int32()
        _check(self.lib.td_model_get_option(self.handle, name.encode(), ctypes.byref(v)), 'td_model_get_option')
        return int(v.value)

    @_device_bound
    def graph_build(self, x: torch.Tensor, mask_ligand: torch.Tensor, node_ptr: torch.Tensor, width: int,
                    max_graph_nodes: int = 0) -> torch.Tensor:
        """The model's graph (its cutoff_mode) on a composed batch as a dense [N, width] table of in-neighbours, -1 padded."""
        N = x.shape[0]
        out = torch.empty(N, width, dtype=torch.int32, device=x.device)
        mask_u8 = mask_ligand.to(torch.uint8).contiguous()
        _check(self.lib.td_graph_build(self.handle, _ptr(x, torch.float32, 'x'), _ptr(mask_u8), _ptr(node_ptr, torch.int32, 'node_ptr'),
                                       N, node_ptr.numel() - 1, max_graph_nodes, _ptr(out), width, _stream(self.device)),
               'td_graph_build')
        return out

    def workspace(self, N: int, B: int, Nl: int) -> torch.Tensor:
        need = int(self.lib.td_workspace_bytes(self.handle, N, B, Nl))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=self.device)
        return self._ws

    @_device_bound
    def graph_ptr(self, batch: torch.Tensor, B: int) -> torch.Tensor:
        ptr = torch.empty(B + 1, dtype=torch.int32, device=batch.device)
        _check(self.lib.td_graph_ptr(_ptr(batch, torch.int64, 'batch'), batch.numel(), B, _ptr(ptr), _stream(self.device)),
               'td_graph_ptr')
        return ptr

    @_device_bound
    def knn(self, x: torch.Tensor, node_ptr: torch.Tensor, k: int = KNN, max_graph_nodes: int = 0) -> torch.Tensor:
        N = x.shape[0]
        out = torch.empty(N, k, dtype=torch.int32, device=x.device)
        _check(self.lib.td_knn(_ptr(x, torch.float32, 'x'), _ptr(node_ptr, torch.int32, 'node_ptr'), N,
                               node_ptr.numel() - 1, k, max_graph_nodes, _ptr(out), _stream(self.device)), 'td_knn')
        return out

    @_device_bound
    def refine_forward(self, h, x, mask_ligand, node_ptr, fix_x=False, max_graph_nodes=0, want_graph=False):
        N, B = h.shape[0], node_ptr.numel() - 1
        out_h = torch.empty_like(h)
        out_x = torch.empty_like(x)
        if want_graph and not self.default_graph:
            raise ValueError('want_graph: the dense [N, 32] graph / gate outputs exist for the k = 32 kNN graph only (graph_build)')
        nbr = torch.empty(N, KNN, dtype=torch.int32, device=h.device) if want_graph else None
        ew = torch.empty(N, KNN, dtype=torch.float32, device=h.device) if want_graph else None
        ws = self.workspace(N, B, 0)
        mask_u8 = mask_ligand.to(torch.uint8).contiguous()
        _check(self.lib.td_refine_forward(
            self.handle, _ptr(h, torch.float32, 'h'), _ptr(x, torch.float32, 'x'), _ptr(mask_u8),
            _ptr(node_ptr, torch.int32, 'node_ptr'), N, B, int(bool(fix_x)), max_graph_nodes, _ptr(out_h), _ptr(out_x),
            _ptr(nbr), _ptr(ew), _ptr(ws), ws.numel(), _stream(self.device)), 'td_refine_forward')
        return out_h, out_x, nbr, ew

    @_device_bound
    def model_forward(self, protein_pos, protein_v, protein_ptr, ligand_pos, ligand_v, ligand_ptr, fix_x=False,
                      max_graph_nodes=0, want_final_h=True, out=None, ligand_graph_bias=None):
        Np, Nl, B = protein_pos.shape[0], ligand_pos.shape[0], protein_ptr.numel() - 1
        N, C = Np + Nl, self.num_classes
        dev = protein_pos.device
        if out is None:
            out = {}
        pred_pos = out.get('pred_ligand_pos')
        if pred_pos is None:
            pred_pos = torch.empty(Nl, 3, dtype=torch.float32, device=dev)
        pred_v = out.get('pred_ligand_v')
        if pred_v is None:
            pred_v = torch.empty(Nl, C, dtype=torch.float32, device=dev)
        lig_h = out.get('final_ligand_h')
        if lig_h is None:
            lig_h = torch.empty(Nl, HIDDEN, dtype=torch.float32, device=dev)
        final_h = torch.empty(N, HIDDEN, dtype=torch.float32, device=dev) if want_final_h else None
        ws = self.workspace(N, B, Nl)
        _check(self.lib.td_model_forward(
            self.handle, _ptr(protein_pos, torch.float32, 'protein_pos'), _ptr(protein_v, torch.float32, 'protein_v'),
            _ptr(protein_ptr, torch.int32, 'protein_ptr'), Np, _ptr(ligand_pos, torch.float32, 'ligand_pos'),
            _ptr(ligand_v, torch.int64, 'ligand_v'), _ptr(ligand_ptr, torch.int32, 'ligand_ptr'), Nl, B,
            int(bool(fix_x)), max_graph_nodes, _ptr(pred_pos), _ptr(pred_v), _ptr(lig_h), _ptr(final_h), _ptr(ws),
            ws.numel(), _graph_bias_ptr(ligand_graph_bias, B), _stream(self.device)), 'td_model_forward')
        return {'pred_ligand_pos': pred_pos, 'pred_ligand_v': pred_v, 'final_h': final_h, 'final_ligand_h': lig_h}

    @_device_bound
    def posterior_step(self, t, ligand_ptr, ligand_pos, ligand_v, pred_pos, pred_v, noise, uniform,
                       pos_next=None, v_next=None, log_v0=None, log_post=None):
        Nl, B = ligand_pos.shape[0], ligand_ptr.numel() - 1
        if pos_next is None:
            pos_next = torch.empty_like(ligand_pos)
        if v_next is None:
            v_next = torch.empty_like(ligand_v)
        _check(self.lib.td_posterior_step(
            self.handle, _ptr(t, torch.int32, 't'), _ptr(ligand_ptr, torch.int32, 'ligand_ptr'), Nl, B,
            _ptr(ligand_pos, torch.float32, 'ligand_pos'), _ptr(ligand_v, torch.int64, 'ligand_v'),
            _ptr(pred_pos, torch.float32, 'pred_pos'), _ptr(pred_v, torch.float32, 'pred_v'),
            _ptr(noise, torch.float32, 'noise'), _ptr(uniform, torch.float32, 'uniform'), _ptr(pos_next),
            _ptr(v_next, torch.int64, 'v_next'), _ptr(log_v0), _ptr(log_post), _stream(self.device)), 'td_posterior_step')
        return pos_next, v_next

    # ---- likelihood estimation / return_all (the other consumers of the denoiser)
    @_device_bound
    def perturb(self, t, ligand_ptr, ligand_pos, ligand_v, noise, uniform):
        Nl, B = ligand_pos.shape[0], ligand_ptr.numel() - 1
        pos_t, v_t = torch.empty_like(ligand_pos), torch.empty_like(ligand_v)
        _check(self.lib.td_perturb(self.handle, _ptr(t, torch.int32, 't'), _ptr(ligand_ptr, torch.int32, 'ligand_ptr'), Nl, B,
                                   _ptr(ligand_pos, torch.float32, 'ligand_pos'), _ptr(ligand_v, torch.int64, 'ligand_v'),
                                   _ptr(noise, torch.float32, 'noise'), _ptr(uniform, torch.float32, 'uniform'),
                                   _ptr(pos_t), _ptr(v_t), _stream(self.device)), 'td_perturb')
        return pos_t, v_t

    @_device_bound
    def likelihood_terms(self, t, ligand_ptr, pos_0, pos_t, v_0, v_t, pred_pos, pred_v):
        Nl, B = pos_0.shape[0], ligand_ptr.numel() - 1
        kl_pos = torch.empty(B, dtype=torch.float32, device=pos_0.device)
        kl_v = torch.empty(B, dtype=torch.float32, device=pos_0.device)
        _check(self.lib.td_likelihood_terms(
            self.handle, _ptr(t, torch.int32, 't'), _ptr(ligand_ptr, torch.int32, 'ligand_ptr'), Nl, B,
            _ptr(pos_0, torch.float32, 'pos_0'), _ptr(pos_t, torch.float32, 'pos_t'), _ptr(v_0, torch.int64, 'v_0'),
            _ptr(v_t, torch.int64, 'v_t'), _ptr(pred_pos, torch.float32, 'pred_pos'), _ptr(pred_v, torch.float32, 'pred_v'),
            _ptr(kl_pos), _ptr(kl_v), _stream(self.device)), 'td_likelihood_terms')
        return kl_pos, kl_v

    @_device_bound
    def likelihood_prior(self, ligand_ptr, pos_0, v_index):
        Nl, B = pos_0.shape[0], ligand_ptr.numel() - 1
        kl_pos = torch.empty(B, dtype=torch.float32, device=pos_0.device)
        kl_v = torch.empty(B, dtype=torch.float32, device=pos_0.device)
        _check(self.lib.td_likelihood_prior(self.handle, _ptr(ligand_ptr, torch.int32, 'ligand_ptr'), Nl, B,
                                            _ptr(pos_0, torch.float32, 'pos_0'), _ptr(v_index, torch.int64, 'v_index'),
                                            _ptr(kl_pos), _ptr(kl_v), _stream(self.device)), 'td_likelihood_prior')
        return kl_pos, kl_v

    @_device_bound
    def embed_ligand(self, ligand_v):
        h = torch.empty(ligand_v.shape[0], HIDDEN, dtype=torch.float32, device=ligand_v.device)
        _check(self.lib.td_embed_ligand(self.handle, _ptr(ligand_v, torch.int64, 'ligand_v'), ligand_v.shape[0], _ptr(h),
                                        _stream(self.device)), 'td_embed_ligand')
        return h

    @_device_bound
    def v_inference(self, h):
        out = torch.empty(h.shape[0], self.num_classes, dtype=torch.float32, device=h.device)
        _check(self.lib.td_v_inference(self.handle, _ptr(h, torch.float32, 'h'), h.shape[0], _ptr(out), _stream(self.device)),
               'td_v_inference')
        return out

    @_device_bound
    def center_pos(self, protein_pos, protein_ptr, ligand_pos, ligand_ptr, offset=None, sign=-1):
        """In place.  offset=None: compute the protein centroids and subtract them (sign=-1)."""
        B = protein_ptr.numel() - 1
        compute = offset is None
        if compute:
            offset = torch.empty(B, 3, dtype=torch.float32, device=protein_ptr.device)
        _check(self.lib.td_center_pos(_ptr(protein_pos) if protein_pos is not None else None,
                                      _ptr(protein_ptr, torch.int32, 'protein_ptr'), _ptr(ligand_pos),
                                      _ptr(ligand_ptr, torch.int32, 'ligand_ptr'), B, _ptr(offset), int(compute), sign,
                                      _stream(self.device)), 'td_center_pos')
        return offset

    @_device_bound
    def debug_node_stage(self, layer: int, stage: int, h: torch.Tensor):
        """Test hook: node projections P [N,512] and query vectors q [N,128] of one attention stage."""
        N = h.shape[0]
        P = torch.empty(N, 4 * HIDDEN, dtype=torch.float32, device=h.device)
        q = torch.empty(N, HIDDEN, dtype=torch.float32, device=h.device)
        _check(self.lib.td_debug_node_stage(self.handle, layer, stage, _ptr(h, torch.float32, 'h'), N, _ptr(P), _ptr(q),
                                            _stream(self.device)), 'td_debug_node_stage')
        return P, q


class NativeSession:
    """Loop-invariant state of one sample_diffusion call (td_session): the centred protein is handed over once."""

    def __init__(self, native: NativeModel, protein_pos, protein_v, protein_ptr, ligand_ptr, num_ligand_atoms: int,
                 max_graph_nodes: int = 0):
        self.native = native
        self.lib = native.lib
        self.Nl = int(num_ligand_atoms)
        self.C = native.num_classes
        self.device = protein_pos.device
        handle = c_void_p()
        with _on(self.device):          # the session block is hipMalloc'ed on the current device
            _check(self.lib.td_session_create(
                native.handle, _ptr(protein_pos, torch.float32, 'protein_pos'), _ptr(protein_v, torch.float32, 'protein_v'),
                _ptr(protein_ptr, torch.int32, 'protein_ptr'), protein_pos.shape[0], _ptr(ligand_ptr, torch.int32, 'ligand_ptr'),
                self.Nl, protein_ptr.numel() - 1, max_graph_nodes, _stream(self.device), ctypes.byref(handle)),
                'td_session_create')
        self.handle = handle

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h and getattr(self, 'lib', None) is not None:
            try:
                with _on(self.device):
                    self.lib.td_session_destroy(h)
            except Exception:            # interpreter shutdown: torch may already be gone
                self.lib.td_session_destroy(h)

    @_device_bound
    def forward(self, ligand_pos, ligand_v, out=None, ligand_graph_bias=None):
        out = out or {}
        dev = ligand_pos.device
        pred_pos = out.get('pred_ligand_pos')
        if pred_pos is None:
            pred_pos = torch.empty(self.Nl, 3, dtype=torch.float32, device=dev)
        pred_v = out.get('pred_ligand_v')
        if pred_v is None:
            pred_v = torch.empty(self.Nl, self.C, dtype=torch.float32, device=dev)
        lig_h = out.get('final_ligand_h')
        if lig_h is None:
            lig_h = torch.empty(self.Nl, HIDDEN, dtype=torch.float32, device=dev)
        _check(self.lib.td_session_forward(self.handle, _ptr(ligand_pos, torch.float32, 'ligand_pos'),
                                           _ptr(ligand_v, torch.int64, 'ligand_v'), _ptr(pred_pos), _ptr(pred_v),
                                           _ptr(lig_h), _graph_bias_ptr(ligand_graph_bias, None), _stream(self.device)),
               'td_session_forward')
        return {'pred_ligand_pos': pred_pos, 'pred_ligand_v': pred_v, 'final_h': None, 'final_ligand_h': lig_h}

    def make_step_io(self, step_index, t_all, ligand_pos, ligand_v, noise, uniform, pos_traj, v_traj, v0_traj=None,
                     vt_traj=None, pos_only=False, ligand_graph_bias=None) -> StepIO:
        """The argument block of :meth:`step`; the tensors must stay alive (and in place) for as long as it is used."""
        S = int(t_all.shape[0])
        if step_index.numel() != 2 or pos_traj.shape[0] != S or v_traj.shape[0] != S:
            raise ValueError('step_index must hold 2 int32, the trajectories one slot per step')
        io = StepIO()
        io.d_step = _ptr(step_index, torch.int32, 'step_index').value
        io.d_t_all = _ptr(t_all, torch.int32, 't_all').value
        io.num_steps, io.pos_only = S, int(bool(pos_only))
        io.d_ligand_pos = _ptr(ligand_pos, torch.float32, 'ligand_pos').value
        io.d_ligand_v = _ptr(ligand_v, torch.int64, 'ligand_v').value
        io.d_noise = _ptr(noise, torch.float32, 'noise').value
        io.d_uniform = _ptr(uniform, torch.float32, 'uniform').value
        io.d_pos_traj = _ptr(pos_traj, torch.float32, 'pos_traj').value
        io.d_v_traj = _ptr(v_traj, torch.int64, 'v_traj').value
        io.d_v0_traj = _ptr(v0_traj, torch.float32, 'v0_traj').value if v0_traj is not None and v0_traj.numel() else None
        io.d_vt_traj = _ptr(vt_traj, torch.float32, 'vt_traj').value if vt_traj is not None and vt_traj.numel() else None
        gb = _graph_bias_ptr(ligand_graph_bias, int(t_all.shape[1]))
        io.d_ligand_graph_bias = gb.value if gb is not None else None
        return io

    def step(self, io: StepIO, use_graph=True):
        """One reverse-diffusion step (denoiser + posterior update + trajectory record) as one replayable unit
        (td_session_step): captured into a hipGraph at the second call, replayed from then on."""
        with _on(self.device):
            _check(self.lib.td_session_step(self.handle, ctypes.byref(io), int(bool(use_graph)), _stream(self.device)),
                   'td_session_step')

    def last_step_was_graph(self) -> bool:
        return bool(self.lib.td_session_step_graph(self.handle))

    def row_counts(self):
        """(N, rows recomputed at layer 0, [receptive-field level sizes ...]) of the last forward; synchronises.

        Level k (1-based) is what the layer k - 1 from the end has to update when only ligand outputs are read."""
        n = self._counts()
        return int(n[0]), int(n[1]), [int(v) for v in n[2:6] if v >= 0]

    def forward_reach_rows(self):
        """Rows layer 1 recomputes (the ligand's one-hop forward reach), or None when that pruning is off."""
        v = int(self._counts()[6])
        return v if v >= 0 else None

    def shared_static_tables(self):
        """(rows of the static tables, distinct protein blocks of the batch) when the session keeps its static tables once per pocket
        (all samples of a pocket carry the same protein block; model option ``session_share_pockets``), else None."""
        n = self._counts()
        return (int(n[7]), int(n[8])) if n[7] >= 0 else None

    @_device_bound
    def _counts(self):
        n = (c_int32 * 10)()
        _check(self.lib.td_session_row_counts(self.handle, n, 10, _stream(self.device)), 'td_session_row_counts')
        return n

    def dirty_rows(self) -> int:
        return self.row_counts()[1]


# ----------------------------------------------------------------------------------------- standalone EGNN refine net
EGNN_LAYER_KEYS = ('edge_mlp.net.0.weight', 'edge_mlp.net.0.bias', 'edge_mlp.net.2.weight', 'edge_mlp.net.2.bias',
                   'edge_inf.0.weight', 'edge_inf.0.bias', 'x_mlp.0.weight', 'x_mlp.0.bias', 'x_mlp.2.weight',
                   'node_mlp.net.0.weight', 'node_mlp.net.0.bias', 'node_mlp.net.2.weight', 'node_mlp.net.2.bias')


def egnn_flat_key_order(num_layers: int):
    return [f'net.{l}.{k}' for l in range(num_layers) for k in EGNN_LAYER_KEYS]


def graph_ptr(batch: torch.Tensor, B: int) -> torch.Tensor:
    """CSR offsets of a sorted PyG ``batch`` vector (td_graph_ptr); no model handle needed."""
    lib = load_library()
    ptr = torch.empty(B + 1, dtype=torch.int32, device=batch.device)
    with _on(batch.device):
        _check(lib.td_graph_ptr(_ptr(batch, torch.int64, 'batch'), batch.numel(), B, _ptr(ptr), _stream(batch.device)),
               'td_graph_ptr')
    return ptr


def protein_centroids(protein_pos: torch.Tensor, protein_ptr: torch.Tensor) -> torch.Tensor:
    """Per-graph centroid [B, 3] of the protein atoms (td_center_pos's offset: one block reduction per graph, so the
    result is run-to-run reproducible, unlike an atomics-based scatter_mean)."""
    lib = load_library()
    B = protein_ptr.numel() - 1
    scratch = protein_pos.detach().clone().contiguous().float()
    offset = torch.empty(B, 3, dtype=torch.float32, device=protein_pos.device)
    with _on(protein_pos.device):
        _check(lib.td_center_pos(_ptr(scratch), _ptr(protein_ptr, torch.int32, 'protein_ptr'), None,
                                 _ptr(protein_ptr, torch.int32, 'protein_ptr'), B, _ptr(offset), 1, -1,
                                 _stream(protein_pos.device)), 'td_center_pos')
    return offset


class NativeEgnn:
    """Owns a td_egnn handle: the EGNN refine net of models/egnn.py with packed weights on the current HIP device."""

    def __init__(self, num_layers: int, state_dict, hidden_dim=HIDDEN, edge_feat_dim=4, k=KNN, device=None, prefix=''):
        self.lib = load_library()
        self.device = _canonical_device(device)
        self.num_layers = int(num_layers)
        blob = np.ascontiguousarray(np.concatenate(
            [state_dict[prefix + key].detach().cpu().numpy().astype(np.float32).reshape(-1)
             for key in egnn_flat_key_order(self.num_layers)]))
        handle = c_void_p()
        with torch.cuda.device(self.device):
            _check(self.lib.td_egnn_create(self.num_layers, hidden_dim, edge_feat_dim, k, blob.ctypes.data_as(POINTER(c_float)),
                                           blob.size, ctypes.byref(handle)), 'td_egnn_create')
        self.handle = handle
        self._ws = None

    def __del__(self):
        h, self.handle = getattr(self, 'handle', None), None
        if h and getattr(self, 'lib', None) is not None:
            self.lib.td_egnn_destroy(h)

    @_device_bound
    def forward(self, h, x, mask_ligand, node_ptr, return_all=False, max_graph_nodes=0):
        N, B, L = h.shape[0], node_ptr.numel() - 1, self.num_layers
        out_h, out_x = torch.empty_like(h), torch.empty_like(x)
        all_h = torch.empty(L, N, HIDDEN, dtype=torch.float32, device=h.device) if return_all else None
        all_x = torch.empty(L, N, 3, dtype=torch.float32, device=h.device) if return_all else None
        need = int(self.lib.td_egnn_workspace_bytes(N))
        if self._ws is None or self._ws.numel() < need:
            self._ws = torch.empty(need, dtype=torch.uint8, device=h.device)
        mask_u8 = mask_ligand.to(torch.uint8).contiguous()
        _check(self.lib.td_egnn_forward(
            self.handle, _ptr(h, torch.float32, 'h'), _ptr(x, torch.float32, 'x'), _ptr(mask_u8),
            _ptr(node_ptr, torch.int32, 'node_ptr'), N, B, max_graph_nodes, _ptr(out_h), _ptr(out_x), _ptr(all_h), _ptr(all_x),
            _ptr(self._ws), self._ws.numel(), _stream(self.device)), 'td_egnn_forward')
        return out_h, out_x, all_h, all_x
