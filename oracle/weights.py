"""Deterministic weights for parity tests and benchmarks (TEST INFRASTRUCTURE).

No pretrained checkpoint ships with the reference (pretrained_models/.gitignore), so parity is against
seeded random weights.  To be independent of module-construction order, every tensor is drawn from its
own generator seeded by (seed, crc32(key)); the same ``state_dict`` loads into the real reference model
(``make_golden.py``), into the restatement and into ``targetdiff_amd.ScorePosNet3D``.

Key names and shapes restate SURVEY.md Appendix C (reference: models/molopt_score_model.py:236-311,
models/uni_transformer.py:26-40,102-106,230-274, models/common.py:60-80); ``make_golden.py`` asserts they
match the real reference ``state_dict`` exactly.
"""
from __future__ import annotations

import zlib
from collections import OrderedDict

import torch

# configs/training.yml:9-42 (model section), the live configuration of the sampler.
DEFAULT_MODEL_CONFIG = dict(
    model_mean_type='C0', beta_schedule='sigmoid', beta_start=1.e-7, beta_end=2.e-3,
    v_beta_schedule='cosine', v_beta_s=0.01, num_diffusion_timesteps=1000, loss_v_weight=100.,
    sample_time_method='symmetric', time_emb_dim=0, time_emb_mode='simple', center_pos_mode='protein',
    node_indicator=True, model_type='uni_o2', num_blocks=1, num_layers=9, hidden_dim=128, n_heads=16,
    edge_feat_dim=4, num_r_gaussian=20, knn=32, num_node_types=8, act_fn='relu', norm=True,
    cutoff_mode='knn', ew_net_type='global', num_x2h=1, num_h2x=1, r_max=10., x2h_out_fc=False,
    sync_twoup=False,
)

PROTEIN_FEATURE_DIM = 27   # utils/transforms.py:115-124 (6 elements + 20 amino acids + backbone flag)
LIGAND_FEATURE_DIM = 13    # utils/transforms.py:48-62 ('add_aromatic' map)


def _mlp_spec(prefix, in_dim, hidden, out_dim):
    return [
        (f'{prefix}.net.0.weight', (hidden, in_dim), 'linear', in_dim),
        (f'{prefix}.net.0.bias', (hidden,), 'bias', in_dim),
        (f'{prefix}.net.1.weight', (hidden,), 'ln_w', 0),
        (f'{prefix}.net.1.bias', (hidden,), 'ln_b', 0),
        (f'{prefix}.net.3.weight', (out_dim, hidden), 'linear', hidden),
        (f'{prefix}.net.3.bias', (out_dim,), 'bias', hidden),
    ]


def _att_layer_spec(prefix, cfg, num_x2h, num_h2x):
    H, heads = cfg['hidden_dim'], cfg['n_heads']
    kv_in = 2 * H + cfg['edge_feat_dim'] + 4 * cfg['num_r_gaussian']
    spec = [(f'{prefix}.distance_expansion.offset', (cfg['num_r_gaussian'],), 'offset', 0)]
    r_feat = 4 * cfg['num_r_gaussian']
    ew = cfg.get('ew_net_type', 'global')
    for i in range(num_x2h):
        p = f'{prefix}.x2h_layers.{i}'
        spec += _mlp_spec(f'{p}.hk_func', kv_in, H, H)
        spec += _mlp_spec(f'{p}.hv_func', kv_in, H, H)
        spec += _mlp_spec(f'{p}.hq_func', H, H, H)
        if ew == 'r':                                   # models/uni_transformer.py:34-35
            spec += [(f'{p}.ew_net.0.weight', (1, r_feat), 'linear', r_feat), (f'{p}.ew_net.0.bias', (1,), 'bias', r_feat)]
        elif ew == 'm':                                 # :36-37
            spec += [(f'{p}.ew_net.0.weight', (1, H), 'linear', H), (f'{p}.ew_net.0.bias', (1,), 'bias', H)]
        if cfg.get('x2h_out_fc', False):                # :39-40
            spec += _mlp_spec(f'{p}.node_output', 2 * H, H, H)
    for i in range(num_h2x):
        p = f'{prefix}.h2x_layers.{i}'
        spec += _mlp_spec(f'{p}.xk_func', kv_in, H, H)
        spec += _mlp_spec(f'{p}.xv_func', kv_in, H, heads)
        spec += _mlp_spec(f'{p}.xq_func', H, H, H)
        if ew == 'r':                                   # :102-103
            spec += [(f'{p}.ew_net.0.weight', (1, r_feat), 'linear', r_feat), (f'{p}.ew_net.0.bias', (1,), 'bias', r_feat)]
    return spec


def parameter_spec(cfg=None, protein_dim=PROTEIN_FEATURE_DIM, ligand_dim=LIGAND_FEATURE_DIM):
    """[(key, shape, kind, fan_in)] for every learnable tensor / fixed offset of the default model."""
    cfg = dict(DEFAULT_MODEL_CONFIG if cfg is None else cfg)
    assert cfg['model_type'] == 'uni_o2' and cfg['time_emb_dim'] == 0
    H = cfg['hidden_dim']
    emb = H - 1 if cfg['node_indicator'] else H
    spec = [
        ('protein_atom_emb.weight', (emb, protein_dim), 'linear', protein_dim),
        ('protein_atom_emb.bias', (emb,), 'bias', protein_dim),
        ('ligand_atom_emb.weight', (emb, ligand_dim), 'linear', ligand_dim),
        ('ligand_atom_emb.bias', (emb,), 'bias', ligand_dim),
        ('refine_net.distance_expansion.offset', (cfg['num_r_gaussian'],), 'offset', 0),
    ]
    if cfg['ew_net_type'] == 'global':                   # models/uni_transformer.py:241-242
        spec += _mlp_spec('refine_net.edge_pred_layer', cfg['num_r_gaussian'], H, 1)
    # init_h_emb_layer: built with num_init_x2h=1, num_init_h2x=0 and never called
    # (models/uni_transformer.py:245,255-261 vs :301-328); present in checkpoints.
    spec += _att_layer_spec('refine_net.init_h_emb_layer', cfg, 1, 0)
    for l in range(cfg['num_layers']):
        spec += _att_layer_spec(f'refine_net.base_block.{l}', cfg, cfg['num_x2h'], cfg['num_h2x'])
    spec += [
        ('v_inference.0.weight', (H, H), 'linear', H),
        ('v_inference.0.bias', (H,), 'bias', H),
        ('v_inference.2.weight', (ligand_dim, H), 'linear', H),
        ('v_inference.2.bias', (ligand_dim,), 'bias', H),
    ]
    return spec


GAUSSIAN_OFFSETS = [0, 1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3, 3.5, 4, 4.5, 5, 5.5, 6, 7, 8, 9, 10]
"""models/common.py:15 (fixed_offset=True)."""


def make_state_dict(seed: int = 2021, cfg=None, protein_dim=PROTEIN_FEATURE_DIM,
                    ligand_dim=LIGAND_FEATURE_DIM, gain: float = 1.0) -> 'OrderedDict[str, torch.Tensor]':
    """Learnable tensors only (load with strict=False: schedule constants/buffers come from the config).

    Linear weights/biases ~ U(-b, b), b = gain/sqrt(fan_in) (nn.Linear's default range when gain=1);
    LayerNorm weight = 1 + 0.2*N(0,1), bias = 0.1*N(0,1) so the affine terms are exercised.
    """
    sd = OrderedDict()
    for key, shape, kind, fan_in in parameter_spec(cfg, protein_dim, ligand_dim):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(key.encode())) % (2 ** 63))
        if kind == 'offset':
            t = torch.tensor(GAUSSIAN_OFFSETS, dtype=torch.float32)
            assert tuple(t.shape) == tuple(shape)
        elif kind in ('linear', 'bias'):
            b = gain / (fan_in ** 0.5)
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) * b
        elif kind == 'ln_w':
            t = 1.0 + 0.2 * torch.randn(shape, generator=g, dtype=torch.float32)
        elif kind == 'ln_b':
            t = 0.1 * torch.randn(shape, generator=g, dtype=torch.float32)
        else:
            raise ValueError(kind)
        sd[key] = t
    return sd


# ------------------------------------------------------------------------------------------ EGNN (models/egnn.py)
def egnn_parameter_spec(num_layers=9, hidden=128, edge_feat_dim=4, num_r_gaussian=1):
    """(key, shape, kind, fan_in) of the EGNN refine net as get_refine_net builds it (models/molopt_score_model.py:34-42:
    num_r_gaussian = 1, so the edge MLP sees [h_i | h_j | d^2 | one_hot(type)] = 2 * hidden + 1 + edge_feat_dim inputs;
    models/egnn.py:22-35)."""
    spec = [('distance_expansion.offset', (len(GAUSSIAN_OFFSETS),), 'offset', 0)]
    ein = 2 * hidden + edge_feat_dim + num_r_gaussian
    for l in range(num_layers):
        p = f'net.{l}'
        spec += [(f'{p}.edge_mlp.net.0.weight', (hidden, ein), 'linear', ein), (f'{p}.edge_mlp.net.0.bias', (hidden,), 'bias', ein),
                 (f'{p}.edge_mlp.net.2.weight', (hidden, hidden), 'linear', hidden), (f'{p}.edge_mlp.net.2.bias', (hidden,), 'bias', hidden),
                 (f'{p}.edge_inf.0.weight', (1, hidden), 'linear', hidden), (f'{p}.edge_inf.0.bias', (1,), 'bias', hidden),
                 (f'{p}.x_mlp.0.weight', (hidden, hidden), 'linear', hidden), (f'{p}.x_mlp.0.bias', (hidden,), 'bias', hidden),
                 (f'{p}.x_mlp.2.weight', (1, hidden), 'linear', hidden),
                 (f'{p}.node_mlp.net.0.weight', (hidden, 2 * hidden), 'linear', 2 * hidden),
                 (f'{p}.node_mlp.net.0.bias', (hidden,), 'bias', 2 * hidden),
                 (f'{p}.node_mlp.net.2.weight', (hidden, hidden), 'linear', hidden),
                 (f'{p}.node_mlp.net.2.bias', (hidden,), 'bias', hidden)]
    return spec


def make_egnn_state_dict(seed: int = 2021, num_layers=9, hidden=128, edge_feat_dim=4):
    """Seeded EGNN weights, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) like nn.Linear (x_mlp.2 too, instead of the reference's
    xavier gain 0.001 initialisation, so that the coordinate update is exercised)."""
    sd = OrderedDict()
    for key, shape, kind, fan_in in egnn_parameter_spec(num_layers, hidden, edge_feat_dim):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(('egnn.' + key).encode())) % (2 ** 63))
        if kind == 'offset':
            t = torch.tensor(GAUSSIAN_OFFSETS, dtype=torch.float32)
        else:
            t = (torch.rand(shape, generator=g, dtype=torch.float32) * 2 - 1) / (fan_in ** 0.5)
        sd[key] = t
    return sd


# ------------------------------------------------------------------------------------------ time embedding (round 4)
def time_emb_state_dict(seed: int, ligand_dim=LIGAND_FEATURE_DIM):
    """make_state_dict(seed) with ``ligand_atom_emb.weight`` widened to [127, ligand_dim + 1] (time_emb_mode = 'simple',
    models/molopt_score_model.py:290-291; the first ligand_dim columns stay make_state_dict's).  'sin' does not run in the reference."""
    sd = make_state_dict(seed)
    w = sd['ligand_atom_emb.weight']
    g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(b'time/ligand_atom_emb.weight')) % (2 ** 63))
    col = (torch.rand((w.shape[0], 1), generator=g, dtype=torch.float32) * 2 - 1) / ((ligand_dim + 1) ** 0.5)
    sd['ligand_atom_emb.weight'] = torch.cat([w, col], dim=1)
    return sd


# ------------------------------------------------------------------------------------------ LayerNorm weights of every sign (round 4)
def ln_signs_state_dict(seed: int):
    """make_state_dict(seed) with the LayerNorm weights of every MLP made adversarial for a fold of the LayerNorm into the neighbouring
    Linears (the product packs the edge MLPs that way, csrc/pack.cpp FoldedMlp): every 7th unit negative, units 5 (mod 31) exactly zero,
    units 3 (mod 29) tiny (1e-3 of their value).  The seeded weights are 1 +- 0.2, i.e. all positive."""
    sd = make_state_dict(seed)
    for key, w in sd.items():
        if key.endswith('.net.1.weight'):
            n = torch.arange(w.numel())
            w = torch.where(n % 7 == 0, -w, w)
            w = torch.where(n % 29 == 3, w * 1e-3, w)
            w = torch.where(n % 31 == 5, torch.zeros_like(w), w)
            sd[key] = w.contiguous()
    return sd


# ------------------------------------------------------------------------------------------ dead LayerNorm units, trained-like sets (round 6)
DEAD_BETAS = (-1.0, 1.0, 3.0)


def ln_dead_state_dict(seed: int):
    """make_state_dict(seed) with, in every MLP, the LayerNorm weight of units 5 (mod 31) exactly 0 and their LayerNorm bias cycling
    through -1, +1, +3 (a pruned / weight-decayed unit whose constant relu(beta) survives), units 3 (mod 29) tiny with a bias of 0.5,
    every 7th unit negative, and the first Linear (net.0 weight and bias) scaled by 4 so that the pre-LayerNorm variance is well above
    one.  The product's LayerNorm fold (csrc/pack.cpp FoldedMlp) divides the bias by |weight|: this is the set that overflowed it."""
    sd = make_state_dict(seed)
    for key in list(sd):
        if key.endswith('.net.1.weight'):
            w, bkey = sd[key], key[:-len('weight')] + 'bias'
            b = sd[bkey].clone()
            n = torch.arange(w.numel())
            w = torch.where(n % 7 == 0, -w, w)
            w = torch.where(n % 29 == 3, w * 1e-3, w)
            b = torch.where(n % 29 == 3, torch.full_like(b, 0.5), b)
            dead = n % 31 == 5
            w = torch.where(dead, torch.zeros_like(w), w)
            cyc = torch.tensor(DEAD_BETAS, dtype=torch.float32)[(n // 31) % len(DEAD_BETAS)]
            b = torch.where(dead, cyc, b)
            sd[key], sd[bkey] = w.contiguous(), b.contiguous()
        elif key.endswith('.net.0.weight') or key.endswith('.net.0.bias'):
            sd[key] = (sd[key] * 4.0).contiguous()
    return sd


def trained_like_state_dict(seed: int, gain: float):
    """A weight regime closer to a trained net than make_state_dict's (no checkpoint ships with the reference): every Linear of the
    MLPs drawn with ``gain`` times nn.Linear's default range (sharper softmaxes, pre-LayerNorm statistics far from unit variance),
    LayerNorm weights log-uniform in [0.05, 5], LayerNorm biases uniform in [-2, 2].  The atom embeddings, v_inference and the output
    Linear of the global edge gate keep make_state_dict's range: inputs and the type head stay in their usual range, and the gate stays
    open (with ``gain`` on its output Linear the sigmoid saturates at 0 on most edges and every message vanishes -- measured: the
    per-layer feature update drops to 1e-3, which tests nothing).  Per-layer updates with this set: |dh| ~ 1-5, |dx| ~ 0.1-1 A."""
    sd = make_state_dict(seed, gain=gain)
    base = make_state_dict(seed)
    for key in list(sd):
        g = torch.Generator().manual_seed((seed * 1000003 + zlib.crc32(('trained/' + key).encode())) % (2 ** 63))
        if key.endswith('.net.1.weight'):
            u = torch.rand(sd[key].shape, generator=g, dtype=torch.float32)
            sd[key] = (0.05 * (100.0 ** u)).contiguous()
        elif key.endswith('.net.1.bias'):
            sd[key] = (torch.rand(sd[key].shape, generator=g, dtype=torch.float32) * 4 - 2).contiguous()
        elif '.net.' not in key or key.startswith('refine_net.edge_pred_layer.net.3.'):
            sd[key] = base[key]
    return sd
