"""Pins oracle/restatement.py against golden vectors produced by the REAL reference model files
(oracle/make_golden.py).  CPU only."""
import numpy as np
import torch

from conftest import load_golden, small_inputs, pocket_1h36
from oracle import restatement as R
from oracle import weights
from targetdiff_amd import workloads


def _maxdiff(a, b):
    return float(np.max(np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))))


def test_schedules_match_reference():
    g = load_golden('schedules.npz')
    s = R.diffusion_schedules()
    for k, v in g.items():
        if k in s:
            np.testing.assert_array_equal(s[k].numpy(), v, err_msg=k)


def test_forward_small_all_stages(state_dict, golden_small):
    g = golden_small
    inp = small_inputs(g)
    col = {}
    out = R.model_forward(state_dict, None, inp['protein_pos'], inp['protein_v'], inp['batch_protein'],
                          inp['ligand_pos'], inp['ligand_v'], inp['batch_ligand'], collect=col)
    # neighbour indices: bit exact (same rule as the shimmed knn_graph the reference ran on)
    np.testing.assert_array_equal(col['nbr'].numpy(), g['nbr'])
    valid = g['nbr'] >= 0
    assert valid.sum(1).min() < 32, 'fixture must contain a graph with fewer than k neighbours'
    assert _maxdiff(col['e_w'].numpy()[valid], g['e_w'][valid]) < 2e-6
    for l in range(9):
        assert _maxdiff(col['h_layers'][l], g['h_layers'][l]) < 2e-5, l
        assert _maxdiff(col['x_layers'][l], g['x_layers'][l]) < 2e-5, l
    assert _maxdiff(out['pred_ligand_pos'], g['pred_ligand_pos']) < 2e-5
    assert _maxdiff(out['pred_ligand_v'], g['pred_ligand_v']) < 2e-5
    assert _maxdiff(out['final_h'], g['final_h']) < 2e-5


def test_forward_small_fix_x(state_dict, golden_small):
    g = load_golden('forward_small_fixx.npz')
    inp = small_inputs(golden_small)
    out = R.model_forward(state_dict, None, inp['protein_pos'], inp['protein_v'], inp['batch_protein'],
                          inp['ligand_pos'], inp['ligand_v'], inp['batch_ligand'], fix_x=True)
    np.testing.assert_array_equal(out['pred_ligand_pos'].numpy(), inp['ligand_pos'].numpy())
    assert _maxdiff(out['pred_ligand_v'], g['pred_ligand_v']) < 2e-5
    assert _maxdiff(out['final_ligand_h'], g['final_ligand_h']) < 2e-5


def test_forward_1h36_real_geometry(state_dict):
    g = load_golden('forward_1h36x2.npz')
    pocket, sizes = pocket_1h36()
    b = workloads.pack_samples(pocket, 2, g['sizes'])
    ppos, lpos, _ = R.center_positions(b.protein_pos, torch.zeros(len(b.ligand_element_batch), 3),
                                       b.protein_element_batch, b.ligand_element_batch)
    col = {}
    out = R.model_forward(state_dict, None, ppos, b.protein_atom_feature.float(), b.protein_element_batch,
                          torch.from_numpy(g['ligand_pos']), torch.from_numpy(g['ligand_v']),
                          b.ligand_element_batch, collect=col)
    np.testing.assert_array_equal(col['nbr'].numpy(), g['nbr'])
    assert _maxdiff(out['pred_ligand_pos'], g['pred_ligand_pos']) < 2e-5
    assert _maxdiff(out['pred_ligand_v'], g['pred_ligand_v']) < 2e-5
    assert _maxdiff(out['final_ligand_h'], g['final_ligand_h']) < 2e-5
    assert _maxdiff(out['final_h'][::16], g['final_h_sample']) < 2e-5


def test_posterior_known_answers():
    g = load_golden('posterior_kat.npz')
    s = R.diffusion_schedules()
    t = torch.from_numpy(g['t'])
    pos_next, v_next, log_v0, log_post = R.posterior_step(
        s, t, torch.from_numpy(g['x_t']), torch.from_numpy(g['v_t']), torch.from_numpy(g['x0']),
        torch.from_numpy(g['v0_logits']), torch.from_numpy(g['batch_ligand']), torch.from_numpy(g['noise']),
        torch.from_numpy(g['uniform']), 13)
    assert _maxdiff(pos_next, g['pos_next']) < 1e-6
    assert _maxdiff(log_v0, g['log_v0']) < 1e-6
    assert _maxdiff(log_post, g['log_post']) < 1e-5
    np.testing.assert_array_equal(v_next.numpy(), g['v_next'])
    # t == 0 graph takes the noiseless branch
    bl = g['batch_ligand']
    c0 = s['posterior_mean_c0_coef'][0].item()
    ct = s['posterior_mean_ct_coef'][0].item()
    np.testing.assert_allclose(pos_next.numpy()[bl == 0], (c0 * g['x0'] + ct * g['x_t'])[bl == 0], atol=1e-6)


def test_sample_diffusion_trajectory(state_dict, golden_small):
    g = load_golden('sample_small.npz')
    from oracle.make_golden import small_batch
    b, lpos, lv = small_batch()
    np.testing.assert_array_equal(lpos.numpy(), g['init_ligand_pos'])
    r = R.sample_diffusion(state_dict, None, b.protein_pos, b.protein_atom_feature.float(),
                           b.protein_element_batch, lpos, lv, b.ligand_element_batch, num_steps=6,
                           noises=torch.from_numpy(g['noises']), uniforms=torch.from_numpy(g['uniforms']),
                           record=True)
    assert _maxdiff(torch.stack(r['pos_traj']), g['pos_traj']) < 5e-5
    np.testing.assert_array_equal(torch.stack(r['v_traj']).numpy(), g['v_traj'])
    assert _maxdiff(torch.stack(r['v0_traj']), g['v0_traj']) < 5e-5
    assert _maxdiff(torch.stack(r['vt_traj']), g['vt_traj']) < 5e-5
    assert _maxdiff(r['pos'], g['pos']) < 5e-5
    np.testing.assert_array_equal(r['v'].numpy(), g['v'])


def test_weight_spec_is_complete():
    spec = weights.parameter_spec()
    n = sum(int(np.prod(s)) for k, s, kind, _ in spec if kind != 'offset')
    assert n == 2824692           # trainable parameters of the default model
    assert len(spec) == 384 - 17  # state-dict entries minus 15 schedule constants and 2 buffers


def test_return_all_matches_reference(state_dict, golden_small):
    g = load_golden('forward_small_return_all.npz')
    inp = small_inputs(golden_small)
    out = R.model_forward(state_dict, None, inp['protein_pos'], inp['protein_v'], inp['batch_protein'],
                          inp['ligand_pos'], inp['ligand_v'], inp['batch_ligand'], return_all=True)
    assert len(out['layer_pred_ligand_pos']) == g['layer_pred_ligand_pos'].shape[0] == 2
    for l in range(2):
        assert _maxdiff(out['layer_pred_ligand_pos'][l], g['layer_pred_ligand_pos'][l]) < 2e-5
        assert _maxdiff(out['layer_pred_ligand_v'][l], g['layer_pred_ligand_v'][l]) < 2e-5


def _likelihood_inputs(g):
    return dict(protein_pos=torch.from_numpy(g['protein_pos']), protein_v=torch.from_numpy(g['protein_feat'].astype(np.float32)),
                batch_protein=torch.from_numpy(g['batch_protein']), ligand_pos=torch.from_numpy(g['ligand_pos']),
                ligand_v=torch.from_numpy(g['ligand_v']), batch_ligand=torch.from_numpy(g['batch_ligand']))


def test_likelihood_estimation_matches_reference(state_dict):
    """scripts/likelihood_est_diffusion.py:30,48 -- KL terms at t = (0, 1, 537) with the reference's recorded draws, and
    the prior term.  Relative tolerance 1e-4 (the t = 0 decoder NLL is O(1e3)), absolute 1e-6 for the near-zero KLs."""
    g = load_golden('likelihood_small.npz')
    inp = _likelihood_inputs(g)
    kl_pos, kl_v = R.likelihood_estimation(state_dict, None, **inp, time_step=torch.from_numpy(g['time_step']),
                                           noise=torch.from_numpy(g['noise']), uniform=torch.from_numpy(g['uniform']))
    np.testing.assert_allclose(kl_pos.numpy(), g['kl_pos'], rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(kl_v.numpy(), g['kl_v'], rtol=1e-4, atol=1e-6)
    klp, klv = R.likelihood_estimation(state_dict, None, **inp, time_step=torch.full((3,), 1000, dtype=torch.long))
    np.testing.assert_allclose(klp.numpy(), g['kl_pos_prior'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(klv.numpy(), g['kl_v_prior'], rtol=1e-4, atol=1e-7)


def test_egnn_matches_reference():
    """Standalone EGNN refine net (models/egnn.py, built as get_refine_net('egnn') does): per-layer coordinates, selected
    per-layer features and the per-layer kNN tables' effect, 9 layers."""
    g = load_golden('egnn_small.npz')
    L = int(g['num_layers'])
    sd = weights.make_egnn_state_dict(2021, num_layers=L)
    col = {}
    out = R.egnn_forward(sd, torch.from_numpy(g['h']), torch.from_numpy(g['x']), torch.from_numpy(g['mask_ligand']),
                         torch.from_numpy(g['batch']), num_layers=L, collect=col)
    for l in range(L):
        assert _maxdiff(col['x_layers'][l], g['all_x'][l + 1]) < 2e-5, l
    assert _maxdiff(col['h_layers'][0], g['h_layer1']) < 2e-5
    assert _maxdiff(col['h_layers'][4], g['h_layer5']) < 5e-5
    assert _maxdiff(out['h'], g['h_final']) < 1e-4
    assert float(np.abs(g['all_x'][-1] - g['all_x'][0]).max()) > 1e-2, 'fixture must move the ligand'
