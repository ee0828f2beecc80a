"""CPU: the oracle restatement against the round-3 fixtures (oracle/make_golden_r3.py): the C5 sweep's graph constructions at
C5 size (the REAL reference's k = 48 / 64 / hybrid forwards on the 1000-atom pocket x 2), reference steps on a hybrid graph
and late-t steps of the reference's 1000-step run on the real pocket.  Sized for about a minute in total."""
import numpy as np
import pytest
import torch

from conftest import load_golden, pocket_1h36
from oracle import draws
from oracle import restatement as R


def _csr_rows(g):
    rp, col = g['row_ptr'], g['col'].astype(np.int64)
    return [sorted(col[rp[i]:rp[i + 1]].tolist()) for i in range(len(rp) - 1)]


def _table_rows(nbr):
    return [sorted(int(j) for j in row if j >= 0) for row in nbr.tolist()]


def _c5_batch():
    from oracle.make_golden_r2 import C5_POCKET, C5_SIZES
    from targetdiff_amd import workloads
    return workloads.pack_samples(workloads.synthetic_pocket(**C5_POCKET), 2, C5_SIZES)


@pytest.mark.parametrize('name,cfg', [('forward_c5_k48', dict(knn=48)), ('forward_c5_hybrid', dict(cutoff_mode='hybrid'))])
def test_restatement_c5_size_general_graphs_vs_reference(state_dict, name, cfg):
    """k = 48 (two chunks per row) and hybrid (six chunks on the 150-atom ligand's rows) at C5 size: the reference's own edge
    sets and outputs.  (k = 64 runs on the GPU side only: the same code path as k = 48, 17 s of CPU.)"""
    from oracle import weights
    g = load_golden(name + '.npz')
    assert str(g['source']) == 'reference'
    b = _c5_batch()
    col = {}
    preds = R.model_forward(state_dict, dict(weights.DEFAULT_MODEL_CONFIG, **cfg), torch.from_numpy(g['protein_pos_centred']),
                            b.protein_atom_feature.float(), b.protein_element_batch, torch.from_numpy(g['ligand_pos']),
                            torch.from_numpy(g['ligand_v'].astype(np.int64)), b.ligand_element_batch, collect=col)
    assert _table_rows(col['nbr']) == _csr_rows(g)
    assert np.max(np.abs(preds['pred_ligand_pos'].numpy() - g['pred_ligand_pos'])) < 2e-5
    assert np.max(np.abs(preds['pred_ligand_v'].numpy() - g['pred_ligand_v'])) < 2e-4
    assert np.max(np.abs(preds['final_ligand_h'].numpy() - g['final_ligand_h'])) < 2e-4
    assert np.max(np.abs(preds['final_h'][::16].numpy() - g['final_h_sample'])) < 2e-4


def _one_step(sd, cfg, batch, pos_in, v_in, t, step, base):
    ppos, lpos, off = R.center_positions(batch.protein_pos, pos_in, batch.protein_element_batch, batch.ligand_element_batch)
    preds = R.model_forward(sd, cfg, ppos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos, v_in,
                            batch.ligand_element_batch)
    sched = R.diffusion_schedules()
    tt = torch.full((batch.num_graphs,), t, dtype=torch.long)
    src = draws.Source(base)
    pos, v, log_v0, log_post = R.posterior_step(sched, tt, lpos, v_in, preds['pred_ligand_pos'], preds['pred_ligand_v'],
                                                batch.ligand_element_batch, src.noise(step, lpos.shape),
                                                src.uniform(step, (lpos.shape[0], 13)), 13)
    return pos + off[batch.ligand_element_batch], v, log_v0, log_post


def test_restatement_hybrid_steps_vs_reference(state_dict):
    """Teacher-forced reverse steps of the reference's own loop on a hybrid graph (1h36 x 2)."""
    from oracle import weights
    from targetdiff_amd import workloads
    g = load_golden('sample_1h36x2_hybrid_20.npz')
    pocket, _ = pocket_1h36()
    batch = workloads.pack_samples(pocket, 2, g['sizes'])
    cfg = dict(weights.DEFAULT_MODEL_CONFIG, cutoff_mode='hybrid')
    for s in (1, 19):
        pos, v, log_v0, _ = _one_step(state_dict, cfg, batch, torch.from_numpy(g['pos_traj'][s - 1]),
                                      torch.from_numpy(g['v_traj'][s - 1].astype(np.int64)), 999 - s, s, int(g['draws_base']))
        assert np.array_equal(v.numpy(), g['v_traj'][s].astype(np.int64)), s
        assert np.max(np.abs(pos.numpy() - g['pos_traj'][s])) < 2e-5, s


def test_restatement_real_pocket_late_steps_vs_reference(state_dict):
    """Late-t steps (t = 1, 0) and a mid-run step of the reference's complete 1000-step run on 1h36 x 2."""
    from targetdiff_amd import workloads
    g = load_golden('sample_1h36x2_1000.npz')
    pocket, _ = pocket_1h36()
    batch = workloads.pack_samples(pocket, 2, g['sizes'])
    kept = {int(s): j for j, s in enumerate(g['kept_steps'])}
    for s in (499, 998, 999):
        pos, v, log_v0, log_post = _one_step(state_dict, None, batch, torch.from_numpy(g['pos_traj'][s - 1]),
                                             torch.from_numpy(g['v_traj'][s - 1].astype(np.int64)), 999 - s, s, int(g['draws_base']))
        assert np.array_equal(v.numpy(), g['v_traj'][s].astype(np.int64)), s
        assert np.max(np.abs(pos.numpy() - g['pos_traj'][s])) < 2e-5, s
        assert np.max(np.abs(log_v0.numpy() - g['v0_traj'][kept[s]])) < 2e-4


@pytest.mark.parametrize('seed,gain', [(7, 1.8), (11, 0.5)])
def test_restatement_other_weight_sets_vs_reference(seed, gain):
    """The small batch under two other seeded weight sets of the real reference (stronger / weaker non-linearity)."""
    from oracle import weights
    from oracle.make_golden import small_batch
    g = load_golden(f'forward_small_seed{seed}.npz')
    b, _, lv = small_batch()
    sd = weights.make_state_dict(seed, gain=gain)
    preds = R.model_forward(sd, None, torch.from_numpy(g['protein_pos_centred']), b.protein_atom_feature.float(), b.protein_element_batch,
                            torch.from_numpy(g['ligand_pos']), lv, b.ligand_element_batch)
    assert np.max(np.abs(preds['pred_ligand_pos'].numpy() - g['pred_ligand_pos'])) < 2e-5
    assert np.max(np.abs(preds['final_h'].numpy() - g['final_h'])) < 2e-4


def test_restatement_layernorm_weights_of_every_sign_vs_reference(golden_small):
    """oracle/make_golden_r4.py forward_ln_signs: LayerNorm weights with negative, zero and tiny entries in every MLP (the all-positive seeded
    weights never exercise the sign handling of the product's LayerNorm fold)."""
    from conftest import load_golden, small_inputs
    from oracle import weights
    from oracle.make_golden import SEED
    g = load_golden('forward_ln_signs.npz')
    inp = small_inputs(golden_small)
    np.testing.assert_array_equal(inp['ligand_pos'].numpy(), g['ligand_pos'])
    out = R.model_forward(weights.ln_signs_state_dict(SEED), None, inp['protein_pos'], inp['protein_v'], inp['batch_protein'],
                          inp['ligand_pos'], inp['ligand_v'], inp['batch_ligand'])
    md = lambda a, b: float(np.max(np.abs(np.asarray(a, np.float64) - np.asarray(b, np.float64))))
    assert md(out['pred_ligand_pos'], g['pred_ligand_pos']) < 2e-5
    assert md(out['pred_ligand_v'], g['pred_ligand_v']) < 2e-5
    assert md(out['final_h'], g['final_h']) < 2e-5
