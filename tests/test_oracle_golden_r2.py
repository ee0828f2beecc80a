"""CPU: the oracle restatement and the portable draws against the round-2 fixtures of the real reference
(oracle/make_golden_r2.py).  Sized for seconds: teacher-forced single steps, not whole trajectories."""
import numpy as np
import torch

from conftest import load_golden, pocket_1h36
from oracle import draws
from oracle import restatement as R


def test_counter_draws_known_answers():
    """The draws are integer hashing only: these values must come out on every platform."""
    u = draws.uniform(7, 3, (4,)).numpy()
    n = draws.normal(7, 3, (2, 3)).numpy()
    assert u.dtype == np.float32 and n.dtype == np.float32
    assert np.all((u >= 0) & (u < 1)) and np.all(np.abs(n) <= 6)
    assert np.array_equal(u * (1 << 24), np.round(u * (1 << 24)))             # multiples of 2^-24
    np.testing.assert_array_equal(u, np.asarray(KAT_U, np.float32))
    np.testing.assert_array_equal(n.ravel(), np.asarray(KAT_N, np.float32))
    big = draws.normal(11, 0, (20000, 3)).numpy()
    assert abs(big.mean()) < 0.02 and abs(big.std() - 1.0) < 0.02
    assert abs(draws.uniform(11, 0, (20000, 13)).numpy().mean() - 0.5) < 0.01


KAT_U = [0.5141875147819519, 0.41352027654647827, 0.3812415599822998, 0.0008453130722045898]
KAT_N = [0.602571964263916, 0.1398754119873047, 0.0869479775428772, -0.017329692840576172, 0.3636655807495117, -0.7140781879425049]


def _one_step(sd, batch, pos_in, v_in, t, step, base):
    ppos, lpos, off = R.center_positions(batch.protein_pos, pos_in, batch.protein_element_batch, batch.ligand_element_batch)
    preds = R.model_forward(sd, None, ppos, batch.protein_atom_feature.float(), batch.protein_element_batch, lpos, v_in,
                            batch.ligand_element_batch)
    sched = R.diffusion_schedules()
    tt = torch.full((batch.num_graphs,), t, dtype=torch.long)
    src = draws.Source(base)
    pos, v, log_v0, log_post = R.posterior_step(sched, tt, lpos, v_in, preds['pred_ligand_pos'], preds['pred_ligand_v'],
                                                batch.ligand_element_batch, src.noise(step, lpos.shape),
                                                src.uniform(step, (lpos.shape[0], 13)), 13)
    return pos + off[batch.ligand_element_batch], v, log_v0, log_post


def test_restatement_late_steps_vs_reference(state_dict):
    """t < 10 (c0[t] -> 1) and the noiseless t = 0 step of the reference's 1000-step run."""
    from oracle.make_golden import small_batch
    g = load_golden('sample_small_1000.npz')
    batch = small_batch()[0]
    kept = {int(s): j for j, s in enumerate(g['kept_steps'])}
    for s in (1, 500, 990, 995, 997, 998, 999):
        pos, v, log_v0, log_post = _one_step(state_dict, batch, torch.from_numpy(g['pos_traj'][s - 1]),
                                             torch.from_numpy(g['v_traj'][s - 1].astype(np.int64)), 999 - s, s, int(g['draws_base']))
        assert np.array_equal(v.numpy(), g['v_traj'][s].astype(np.int64)), s
        assert np.max(np.abs(pos.numpy() - g['pos_traj'][s])) < 2e-5, s
        if s in kept:
            assert np.max(np.abs(log_v0.numpy() - g['v0_traj'][kept[s]])) < 2e-4
            assert np.max(np.abs(np.exp(log_post.numpy().astype(np.float64)) - np.exp(g['vt_traj'][kept[s]].astype(np.float64)))) < 1e-6


def test_restatement_c1_steps_vs_reference(state_dict):
    from targetdiff_amd import workloads
    g = load_golden('c1_full.npz')
    pocket, _ = pocket_1h36()
    batch = workloads.pack_samples(pocket, 4, g['sizes'])
    for s in (50, 99):
        pos, v, _, _ = _one_step(state_dict, batch, torch.from_numpy(g['pos_traj'][s - 1]),
                                 torch.from_numpy(g['v_traj'][s - 1].astype(np.int64)), 999 - s, s, int(g['draws_base']))
        assert np.array_equal(v.numpy(), g['v_traj'][s].astype(np.int64))
        assert np.max(np.abs(pos.numpy() - g['pos_traj'][s])) < 2e-5


def test_restatement_forward_c5_shape(state_dict):
    from oracle.make_golden_r2 import C5_POCKET, C5_SIZES
    from targetdiff_amd import workloads
    g = load_golden('forward_c5.npz')
    b = workloads.pack_samples(workloads.synthetic_pocket(**C5_POCKET), 2, C5_SIZES)
    col = {}
    preds = R.model_forward(state_dict, None, torch.from_numpy(g['protein_pos_centred']), b.protein_atom_feature.float(),
                            b.protein_element_batch, torch.from_numpy(g['ligand_pos']),
                            torch.from_numpy(g['ligand_v'].astype(np.int64)), b.ligand_element_batch, collect=col)
    assert np.array_equal(col['nbr'].numpy(), g['nbr'].astype(np.int64))
    assert np.max(np.abs(preds['pred_ligand_pos'].numpy() - g['pred_ligand_pos'])) < 2e-5
    assert np.max(np.abs(preds['pred_ligand_v'].numpy() - g['pred_ligand_v'])) < 2e-4
    assert np.max(np.abs(preds['final_ligand_h'].numpy() - g['final_ligand_h'])) < 2e-4


# ------------------------------------------------------------------------------------------ other graph constructions
def _csr_rows(g):
    rp, col = g['row_ptr'], g['col']
    return [sorted(col[rp[i]:rp[i + 1]].tolist()) for i in range(len(rp) - 1)]


def _table_rows(nbr):
    return [sorted(int(j) for j in row if j >= 0) for row in nbr.tolist()]


def _variant_batch(name):
    from oracle.make_golden import small_batch
    from oracle.make_golden_r2 import hybrid_small_batch
    from targetdiff_amd import workloads
    if name == 'forward_1h36x2_hybrid':
        pocket, sizes = pocket_1h36()
        return workloads.pack_samples(pocket, 2, sizes[:2])
    return (hybrid_small_batch if 'hybrid' in name else small_batch)()[0]


import pytest


@pytest.mark.parametrize('name,cfg', [('forward_small_k16', dict(knn=16)), ('forward_small_k48', dict(knn=48)),
                                      ('forward_small_k64', dict(knn=64)), ('forward_small_hybrid', dict(cutoff_mode='hybrid')),
                                      ('forward_1h36x2_hybrid', dict(cutoff_mode='hybrid'))])
def test_restatement_other_graphs_vs_reference(state_dict, name, cfg):
    """k-NN with k != 32 and cutoff_mode='hybrid' (models/uni_transformer.py:276-286, models/common.py:165-212): the
    reference's own edge lists (as neighbour sets) and outputs."""
    from oracle import weights
    g = load_golden(name + '.npz')
    b = _variant_batch(name)
    full = dict(weights.DEFAULT_MODEL_CONFIG, **cfg)
    # the fixtures hold centred ligand positions; centre the protein the same way
    ppos, _, _ = R.center_positions(b.protein_pos, torch.zeros(b.ligand_element_batch.numel(), 3), b.protein_element_batch,
                                    b.ligand_element_batch)
    col = {}
    preds = R.model_forward(state_dict, full, ppos, b.protein_atom_feature.float(), b.protein_element_batch,
                            torch.from_numpy(g['ligand_pos']), torch.from_numpy(g['ligand_v'].astype(np.int64)),
                            b.ligand_element_batch, collect=col)
    assert _table_rows(col['nbr']) == _csr_rows(g)
    assert np.max(np.abs(preds['pred_ligand_pos'].numpy() - g['pred_ligand_pos'])) < 2e-5
    assert np.max(np.abs(preds['pred_ligand_v'].numpy() - g['pred_ligand_v'])) < 2e-4
    assert np.max(np.abs(preds['final_ligand_h'].numpy() - g['final_ligand_h'])) < 2e-4
    assert np.max(np.abs(torch.stack(col['x_layers']).numpy() - g['x_layers'])) < 2e-5


def test_radius_rule_known_answers():
    """The project's radius-with-cap rule on a hand-checkable case."""
    from oracle import shims
    x = torch.tensor([[0., 0, 0], [1, 0, 0], [2, 0, 0], [3, 0, 0], [0.5, 0, 0], [10, 0, 0], [10.5, 0, 0]])
    batch = torch.tensor([0, 0, 0, 0, 0, 1, 1])
    t = shims.radius_neighbours(x, 1.5, batch, 2)
    assert t.tolist() == [[1, 4], [0, 2], [1, 3], [2, -1], [0, 1], [6, -1], [5, -1]]
    t = shims.radius_neighbours(x, 1.0, batch, 4)             # strict: d = 1.0 is outside
    assert t[0].tolist() == [4, -1, -1, -1] and t[1].tolist() == [4, -1, -1, -1] and t[5].tolist() == [6, -1, -1, -1]
