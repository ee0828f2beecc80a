"""The HIP path in weight regimes the seeded U(-1/sqrt(fan_in), 1/sqrt(fan_in)) sets never reach, against fixtures the REAL reference
produced (oracle/make_golden_r6.py; no checkpoint ships with the reference): LayerNorm units that are exactly dead (weight 0, bias
-1 / +1 / +3) beside tiny and negative ones with the first Linear at x 4, and two "trained-like" sets (every MLP Linear at 4 / 8 times
nn.Linear's range, LayerNorm weights log-uniform in [0.05, 5], biases in [-2, 2]).  Needs an MI355X: ``-m gpu``; every call goes through
the C ABI.

Tolerances, stated here.  Dead units and gain 4: the single-forward gate of tests/_tol.py (5e-6) -- except where the fp32 REFERENCE
itself is further than that from its own float64 evaluation (the fixtures hold both): a forward output is held to
max(5e-6, 2 x |reference fp32 - reference float64|), which is 8e-6 for gain 4 and 3.2e-4 A / 3.8e-4 for gain 8 (|h| reaches 13 and
one forward moves atoms by 4.6 A there: two correct fp32 evaluations differ by that much, the CPU restatement does too), and the HIP
result must sit as close to the float64 values as the fp32 reference does (within a factor 2).  Teacher-forced single steps: sampled
types exact, |dx| <= 1e-5 A, log-probabilities 1e-4 (gain 8: 2 x the reference's own float64 distance on pred_ligand_v, 2.7e-5 -> 1e-4)."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
from _tol import TOL_X, TOL_H, TOL_FWD, close, maxdiff as _maxdiff
from test_oracle_golden_r6 import regime_state_dict, regime_tolerance


def _dev():
    if not torch.cuda.is_available():
        pytest.skip('no HIP device')
    return torch.device('cuda:0')


def _model(sd):
    from oracle import weights
    from targetdiff_amd.models import ScorePosNet3D
    m = ScorePosNet3D(dict(weights.DEFAULT_MODEL_CONFIG), 27, 13)
    assert not m.load_state_dict(sd, strict=False).unexpected_keys
    return m.to(_dev()).eval()


def _args(g, dev):
    from oracle.make_golden import small_batch
    b = small_batch()[0].to(dev)
    return (torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
            torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch)


@pytest.mark.parametrize('name', ['forward_ln_dead.npz', 'forward_trained_g4.npz', 'forward_trained_g8.npz'])
def test_forward_weight_regimes_vs_reference(name):
    """One forward (return_all) per variant of the attention passes' arithmetic: first layer on f16 piece pairs (default), on the exact bf16
    piece triples or on fp32, second layer on f16 piece pairs (default) or fp32 -- every combination is held to the same tolerance against
    the same golden."""
    dev = _dev()
    g = load_golden(name)
    sd = regime_state_dict(name)
    seen = {}
    for split, l1, l2 in ((1, 1, 1), (1, 0, 1), (1, 0, 0), (0, 0, 0)):
        model = _model(sd)
        assert model._native(dev).get_option('edge_first_layer_f16') == 1 and model._native(dev).get_option('edge_second_layer_f16') == 1     # shipped defaults
        model._native(dev).set_option('edge_key_split', split)
        model._native(dev).set_option('edge_first_layer_f16', l1)
        model._native(dev).set_option('edge_second_layer_f16', l2)
        assert model._native(dev).get_option('edge_second_layer_f16') == l2 and model._native(dev).get_option('edge_first_layer_f16') == l1
        p = model(*_args(g, dev), return_all=True)
        for k in ('pred_ligand_pos', 'pred_ligand_v', 'final_h'):
            close(p[k], g[k], regime_tolerance(g, k, TOL_FWD), (name, 'split', split, 'f16 first layer', l1, 'f16 second layer', l2, k))
            d64, r64 = _maxdiff(p[k], g[k + '_f64']), _maxdiff(g[k], g[k + '_f64'])
            print(f'{name} first layer split {split} / f16 {l1}, second layer f16 {l2}, {k}: HIP vs float64 {d64:.3e}, fp32 reference vs float64 {r64:.3e}')
            assert d64 <= max(TOL_FWD, 2.0 * r64), (name, k, d64, r64)
        seen[(split, l1, l2)] = p['final_h'].clone()
        tol_h = regime_tolerance(g, 'final_h', TOL_FWD)
        close(p['final_ligand_h'], g['final_ligand_h'], tol_h, (name, 'final_ligand_h'))
        close(p['layer_pred_ligand_v'][0], g['layer0_pred_ligand_v'], regime_tolerance(g, 'pred_ligand_v', TOL_FWD), (name, 'layer 0 v'))
        close(p['layer_pred_ligand_pos'][0], g['layer0_pred_ligand_pos'], regime_tolerance(g, 'pred_ligand_pos', TOL_FWD), (name, 'layer 0 pos'))
    assert not torch.equal(seen[(1, 1, 1)], seen[(1, 0, 1)]) and not torch.equal(seen[(1, 0, 1)], seen[(1, 0, 0)]) and not torch.equal(seen[(1, 0, 0)], seen[(0, 0, 0)])          # every option is live


def test_dead_units_used_to_overflow_and_other_weights_differ():
    """The fixture is the one the round-5 fold overflowed on (rsqrt(inf) = 0 collapsed every activation of an edge to its bias term);
    and the answer depends on the dead units' constants: with their biases zeroed the outputs move."""
    dev = _dev()
    g = load_golden('forward_ln_dead.npz')
    sd = regime_state_dict('forward_ln_dead.npz')
    p = _model(sd)(*_args(g, dev))
    close(p['final_h'], g['final_h'], TOL_FWD)
    sd2 = {k: v.clone() for k, v in sd.items()}
    for k in sd2:
        if k.endswith('.net.1.weight'):
            sd2[k[:-6] + 'bias'][sd2[k] == 0] = 0.0
    assert _maxdiff(_model(sd2)(*_args(g, dev))['final_h'], g['final_h']) > 1e-2


def test_fold_refuses_what_it_cannot_represent():
    """A live unit (|weight| above 2^-30 of the MLP's largest) whose bias / |weight| would push the folded scale M past 1e15 cannot be
    packed without overflowing fp32 in the kernels' variance: td_model_create refuses instead of computing something else."""
    from oracle import weights
    from oracle.make_golden import SEED
    dev = _dev()
    sd = weights.make_state_dict(SEED)
    k = 'refine_net.base_block.2.x2h_layers.0.hv_func.net.1.'
    sd[k + 'weight'][17] = 1e-8
    sd[k + 'bias'][17] = 1e8
    model = _model(sd)
    with pytest.raises(RuntimeError, match='LayerNorm'):
        model._native(dev)
    sd[k + 'weight'][17] = 1e-10          # below the floor: a dead unit, the constant relu(1e8) goes into the second Linear's bias
    assert _model(sd)._native(dev) is not None


@pytest.mark.parametrize('gain', [4, 8])
def test_teacher_forced_steps_trained_like_vs_reference(gain):
    """20 reverse steps of the reference's own loop with the trained-like weights; every step restarted from the reference's recorded state."""
    from oracle import draws, weights
    from oracle.make_golden import SEED, small_batch
    from test_gpu_long_parity import _one_step
    dev = _dev()
    g = load_golden(f'sample_trained_g{gain}_20.npz')
    model = _model(weights.trained_like_state_dict(SEED, float(gain)))
    batch = small_batch()[0]
    steps, base = int(g['steps']), int(g['draws_base'])
    worst_x = worst_v = 0.0
    for s in range(steps):
        pos_in = torch.from_numpy(g['init_ligand_pos'] if s == 0 else g['pos_traj'][s - 1])
        v_in = torch.from_numpy((g['init_ligand_v'] if s == 0 else g['v_traj'][s - 1]).astype(np.int64))
        pos, v, log_v0, _ = _one_step(model, batch, pos_in, v_in, 999 - s, s, base, dev)
        assert np.array_equal(v.cpu().numpy(), g['v_traj'][s].astype(np.int64)), f'gain {gain}: types differ at step {s}'
        worst_x = max(worst_x, close(pos, g['pos_traj'][s], TOL_X, (gain, 'step', s)))
        worst_v = max(worst_v, close(log_v0, g['v0_traj'][s], TOL_H, (gain, 'v0', s)))
    print(f'gain {gain}: teacher-forced {steps} steps, max |dx| = {worst_x:.3e}, max |d log v0| = {worst_v:.3e}')
    # free-running through the session for the same 20 steps: types equal, positions within the trajectory gate
    from _tol import TOL_TRAJ
    b = batch.to(dev)
    r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                               torch.from_numpy(g['init_ligand_pos']).to(dev), torch.from_numpy(g['init_ligand_v']).to(dev),
                               b.ligand_element_batch, num_steps=steps, center_pos_mode='protein', noise_source=draws.Source(base, dev))
    assert np.array_equal(torch.stack(r['v_traj']).cpu().numpy(), g['v_traj'].astype(np.int64))
    close(torch.stack(r['pos_traj']), g['pos_traj'], TOL_TRAJ, (gain, 'free run'))
