"""The oracle restatement against the round-6 fixtures of the real reference (oracle/make_golden_r6.py): dead LayerNorm units and the two
trained-like weight regimes.  CPU only.

Tolerance: the fp32 reference itself differs from its own float64 evaluation (``*_f64`` in the fixtures) by 3e-7 (dead units), 4e-6 (gain 4)
and 2e-4 (gain 8) -- the features reach |h| = 13 and the coordinates move by 4.6 A in one forward at gain 8 -- so a second fp32 implementation is held
to max(2e-5, 2 x that distance), and must be as close to the float64 values as the reference is (within a factor 2)."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import restatement as R
from oracle import weights
from oracle.make_golden import SEED, small_batch

KEYS = ('pred_ligand_pos', 'pred_ligand_v', 'final_h')


def regime_state_dict(name):
    if name == 'forward_ln_dead.npz':
        return weights.ln_dead_state_dict(SEED)
    return weights.trained_like_state_dict(SEED, 4.0 if 'g4' in name else 8.0)


def md(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    return float(np.max(np.abs(a - np.asarray(b, np.float64))))


def regime_tolerance(g, key, floor):
    """max(floor, 2 x |fp32 reference - float64 reference|) for one output of a round-6 forward fixture"""
    return max(floor, 2.0 * md(g[key], g[key + '_f64']))


@pytest.mark.parametrize('name', ['forward_ln_dead.npz', 'forward_trained_g4.npz', 'forward_trained_g8.npz'])
def test_restatement_weight_regimes_vs_reference(name):
    g = load_golden(name)
    b = small_batch()[0]
    out = R.model_forward(regime_state_dict(name), None, torch.from_numpy(g['protein_pos']), b.protein_atom_feature.float(),
                          b.protein_element_batch, torch.from_numpy(g['ligand_pos']), torch.from_numpy(g['ligand_v']), b.ligand_element_batch)
    for k in KEYS:
        d, tol = md(out[k], g[k]), regime_tolerance(g, k, 2e-5)
        print(name, k, f'{d:.3e} (tolerance {tol:.1e}; restatement vs float64 {md(out[k], g[k + "_f64"]):.3e}, reference vs float64 {md(g[k], g[k + "_f64"]):.3e})')
        assert d <= tol, (name, k, d, tol)
        assert md(out[k], g[k + '_f64']) <= max(2e-5, 2.0 * md(g[k], g[k + '_f64'])), (name, k)


def test_dead_unit_fixture_is_adversarial():
    sd = weights.ln_dead_state_dict(SEED)
    w = sd['refine_net.base_block.3.x2h_layers.0.hv_func.net.1.weight']
    bb = sd['refine_net.base_block.3.x2h_layers.0.hv_func.net.1.bias']
    dead = w == 0
    assert int(dead.sum()) == 4 and sorted(set(bb[dead].tolist())) == [-1.0, 1.0, 3.0]
    assert int((w < 0).sum()) >= 15 and int(((w.abs() < 5e-3) & ~dead).sum()) >= 3
    t = weights.trained_like_state_dict(SEED, 8.0)['refine_net.base_block.3.x2h_layers.0.hv_func.net.1.weight']
    assert float(t.min()) >= 0.05 and float(t.max()) <= 5.0 and float(t.max() / t.min()) > 30
