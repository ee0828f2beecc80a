"""Configurations of ScorePosNet3D outside configs/training.yml that the mirror accepts since round 4, against fixtures the REAL reference
produced (oracle/make_golden_r4.py): the time embedding (time_emb_mode = 'simple': one more input column of ligand_atom_emb, fed with
t / T per graph; 'sin' is dead code in the reference and refused) and model_mean_type = 'noise' (the network's position output is
x_t + eps).  Every step form: the captured graph, launch by launch, and the stateless forward."""
import numpy as np
import pytest
import torch

from conftest import load_golden

pytestmark = pytest.mark.gpu
from _tol import TOL_X, TOL_H, TOL_FWD, TOL_TRAJ, close, maxdiff as _maxdiff


def _dev():
    return torch.device('cuda:0')


def _model(sd, **over):
    from oracle import weights
    from targetdiff_amd.models import ScorePosNet3D
    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    cfg.update(over)
    m = ScorePosNet3D(cfg, 27, 13)
    res = m.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    return m.to(_dev()).eval()


def test_forward_with_simple_time_embedding_vs_reference():
    from oracle import weights
    from oracle.make_golden import SEED, small_batch
    dev = _dev()
    g = load_golden('forward_time_simple.npz')
    model = _model(weights.time_emb_state_dict(SEED), time_emb_dim=int(g['time_emb_dim']), time_emb_mode='simple')
    assert model.ligand_atom_emb.weight.shape == (127, 14)
    b = small_batch()[0].to(dev)
    t = torch.from_numpy(g['time_step']).to(dev)
    p = model(torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
              torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch, time_step=t,
              return_all=True)
    close(p['pred_ligand_pos'], g['pred_ligand_pos'], TOL_X)
    close(p['pred_ligand_v'], g['pred_ligand_v'], TOL_H)
    close(p['final_ligand_h'], g['final_ligand_h'], TOL_H)
    close(p['layer_pred_ligand_v'][0], g['layer0_pred_ligand_v'], TOL_H)
    # the time step matters (another one moves the outputs), and it is required
    p2 = model(torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
               torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch, time_step=t * 0)
    assert _maxdiff(p2['pred_ligand_v'], g['pred_ligand_v']) > 1e-3
    with pytest.raises(ValueError, match='time_step'):
        model(torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
              torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch)


@pytest.mark.parametrize('fixture,over', [('sample_time_simple_6.npz', dict(time_emb_dim=8, time_emb_mode='simple')),
                                          ('sample_noise_6.npz', dict(model_mean_type='noise'))])
def test_sampling_with_time_embedding_and_noise_mean_type_vs_reference(fixture, over):
    from oracle import draws, weights
    from oracle.make_golden import SEED, small_batch
    dev = _dev()
    g = load_golden(fixture)
    sd = weights.time_emb_state_dict(SEED) if 'time_emb_dim' in over else weights.make_state_dict(SEED)
    model = _model(sd, **over)
    b = small_batch()[0].to(dev)
    steps = int(g['steps'])
    side = torch.cuda.Stream(device=dev)
    outs = []
    for kw in (dict(use_graph=True), dict(use_graph=False), dict(use_session=False)):
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                                       torch.from_numpy(g['init_ligand_pos']).to(dev), torch.from_numpy(g['init_ligand_v']).to(dev),
                                       b.ligand_element_batch, num_steps=steps, center_pos_mode='protein',
                                       noise_source=draws.Source(int(g['draws_base']), dev), **kw)
        torch.cuda.current_stream(dev).wait_stream(side)
        assert np.array_equal(torch.stack(r['v_traj']).numpy(), g['v_traj'].astype(np.int64)), (fixture, kw)
        close(torch.stack(r['pos_traj']), g['pos_traj'], TOL_TRAJ, (fixture, kw))
        close(torch.stack(r['v0_traj']), g['v0_traj'], TOL_H, (fixture, kw))
        outs.append(r)
    for r in outs[1:]:
        for key in ('pos_traj', 'v_traj', 'v0_traj', 'vt_traj'):
            assert torch.equal(torch.stack(outs[0][key]), torch.stack(r[key])), (fixture, key)


def test_two_blocks_vs_reference():
    """num_blocks = 2 (models/uni_transformer.py:306-323): the nine layers applied twice with the graph and the edge gate rebuilt from the
    moved coordinates in between.  Forward (and fix_x) on the small batch, 4 reverse steps -- session (which does not cache with several
    blocks), launch by launch, stateless -- and the hybrid graph for the chunked path (self-consistency: session == stateless)."""
    from oracle import draws, weights
    from oracle.make_golden import SEED, small_batch
    dev = _dev()
    g = load_golden('forward_blocks2.npz')
    sd = weights.make_state_dict(SEED)
    model = _model(sd, num_blocks=2)
    b = small_batch()[0].to(dev)
    args = (torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
            torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch)
    p = model(*args)
    close(p['pred_ligand_pos'], g['pred_ligand_pos'], TOL_X)
    close(p['pred_ligand_v'], g['pred_ligand_v'], TOL_H)
    close(p['final_ligand_h'], g['final_ligand_h'], TOL_H)
    close(p['final_h'], g['final_h'], TOL_H)
    f = model(*args, fix_x=True)
    close(f['final_ligand_h'], g['fix_x_final_ligand_h'], TOL_H)
    with pytest.raises(NotImplementedError, match='return_all'):
        model(*args, return_all=True)
    one = _model(sd)(*args)                       # one block gives something else
    assert _maxdiff(one['pred_ligand_v'], g['pred_ligand_v']) > 1e-3
    gs = load_golden('sample_blocks2_4.npz')
    outs = []
    for kw in (dict(use_graph=False), dict(use_session=False)):
        r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                                   torch.from_numpy(gs['init_ligand_pos']).to(dev), torch.from_numpy(gs['init_ligand_v']).to(dev),
                                   b.ligand_element_batch, num_steps=int(gs['steps']), center_pos_mode='protein',
                                   noise_source=draws.Source(int(gs['draws_base']), dev), **kw)
        assert np.array_equal(torch.stack(r['v_traj']).numpy(), gs['v_traj'].astype(np.int64)), kw
        close(torch.stack(r['pos_traj']), gs['pos_traj'], TOL_TRAJ, kw)
        outs.append(r)
    assert torch.equal(torch.stack(outs[0]['pos_traj']), torch.stack(outs[1]['pos_traj']))
    hyb = _model(sd, num_blocks=2, cutoff_mode='hybrid')
    rs = [hyb.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                               torch.from_numpy(gs['init_ligand_pos']).to(dev), torch.from_numpy(gs['init_ligand_v']).to(dev),
                               b.ligand_element_batch, num_steps=3, center_pos_mode='protein',
                               noise_source=draws.Source(4600, dev), use_session=us) for us in (True, False)]
    assert torch.equal(torch.stack(rs[0]['pos_traj']), torch.stack(rs[1]['pos_traj']))


def test_layernorm_weights_of_every_sign_vs_reference():
    """The edge MLPs' LayerNorm is folded into their Linears at pack time (csrc/pack.cpp FoldedMlp: centred first Linear, the sign of the
    LayerNorm weight in its rows, |weight| x the folded scale M in the second Linear's columns, the ReLU as the FMA's clamp: nothing per edge for the consumer).  The seeded weights are all
    positive; this fixture of the real reference (oracle/make_golden_r4.py) has negative, zero and tiny LayerNorm weights in every MLP.
    Both first-layer variants (bf16 piece triples, fp32), the stateless forward and a session's."""
    from oracle import weights
    from oracle.make_golden import SEED, small_batch
    dev = _dev()
    g = load_golden('forward_ln_signs.npz')
    sd = weights.ln_signs_state_dict(SEED)
    lnw = sd['refine_net.base_block.0.x2h_layers.0.hk_func.net.1.weight']
    assert (lnw < 0).sum() >= 15 and (lnw == 0).sum() >= 4 and ((lnw.abs() < 5e-3) & (lnw != 0)).sum() >= 3
    b = small_batch()[0].to(dev)
    args = (torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
            torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch)
    for split in (1, 0):
        model = _model(sd)
        nat = model._native(dev)
        nat.set_option('edge_key_split', split)
        p = model(*args, return_all=True)
        d = {k: _maxdiff(p[k], g[k]) for k in ('pred_ligand_pos', 'pred_ligand_v', 'final_ligand_h', 'final_h')}
        print('edge_key_split', split, d)
        assert d['pred_ligand_pos'] <= TOL_X and d['pred_ligand_v'] <= TOL_H and d['final_ligand_h'] <= TOL_H and d['final_h'] <= TOL_H
        close(p['layer_pred_ligand_v'][0], g['layer0_pred_ligand_v'], TOL_H)
        close(p['layer_pred_ligand_pos'][0], g['layer0_pred_ligand_pos'], TOL_X)
    # the all-positive weights give something else
    one = _model(weights.make_state_dict(SEED))(*args)
    assert _maxdiff(one['pred_ligand_v'], g['pred_ligand_v']) > 1e-3


@pytest.mark.parametrize('name,over', [('forward_ew_r_out_fc.npz', dict(ew_net_type='r', x2h_out_fc=True)),
                                       ('forward_ew_none.npz', dict(ew_net_type='none')),
                                       ('forward_out_fc.npz', dict(x2h_out_fc=True)),
                                       ('forward_ew_m.npz', dict(ew_net_type='m')),
                                       ('forward_sync_twoup.npz', dict(sync_twoup=True)),
                                       ('forward_stages_2_2.npz', dict(num_x2h=2, num_h2x=2)),
                                       ('forward_stages_1_3_r.npz', dict(num_x2h=1, num_h2x=3, ew_net_type='r', x2h_out_fc=True))])
def test_gate_and_output_options_vs_reference(name, over):
    """ew_net_type = 'r' (every stage's own gate on the layer's radial features), 'm' (the x2h gate from the edge's value vector, which the
    kernels never form: its logit is a dot product with the hidden activations), any value that means e_w = 1, and x2h_out_fc = True
    (node_output([attention output | h]) + h) -- 'r' + out_fc are the reference CLASS's defaults (models/uni_transformer.py:146-148,
    212-214), configs/training.yml uses 'global' / False.  Forward and fix_x against fixtures of the real reference; strict state_dict."""
    from oracle import weights
    from oracle.make_golden import SEED, small_batch
    dev = _dev()
    g = load_golden(name)
    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    cfg.update(over)
    sd = weights.make_state_dict(SEED, cfg)
    model = _model(sd, **over)
    learnable = {k for k, p in model.named_parameters() if p.requires_grad and not k.startswith('refine_net.init_h_emb_layer')}
    assert learnable <= set(sd), sorted(learnable - set(sd))[:4]          # every parameter of the mirror came from the reference's keys
    assert ('refine_net.edge_pred_layer.net.0.weight' in dict(model.named_parameters())) == (over.get('ew_net_type', 'global') == 'global')
    b = small_batch()[0].to(dev)
    args = (torch.from_numpy(g['protein_pos']).to(dev), b.protein_atom_feature.float(), b.protein_element_batch,
            torch.from_numpy(g['ligand_pos']).to(dev), torch.from_numpy(g['ligand_v']).to(dev), b.ligand_element_batch)
    p = model(*args)
    d = {k: _maxdiff(p[k], g[k]) for k in ('pred_ligand_pos', 'pred_ligand_v', 'final_ligand_h', 'final_h')}
    print(name, d)
    assert d['pred_ligand_pos'] <= TOL_X and d['pred_ligand_v'] <= TOL_H and d['final_ligand_h'] <= TOL_H and d['final_h'] <= TOL_H
    f = model(*args, fix_x=True)
    close(f['final_ligand_h'], g['fix_x_final_ligand_h'], TOL_H)
    assert torch.equal(f['pred_ligand_pos'].cpu(), torch.from_numpy(g['ligand_pos']))
    base = _model(weights.make_state_dict(SEED))(*args)            # configs/training.yml's options give something else
    assert max(_maxdiff(base['pred_ligand_v'], g['pred_ligand_v']), _maxdiff(base['pred_ligand_pos'], g['pred_ligand_pos'])) > 10 * TOL_X


def test_gate_and_output_options_sampling_vs_reference():
    """4 reverse steps of the reference's loop with ew_net_type = 'r' and x2h_out_fc = True: the session (which does not cache with these
    options), launch by launch, and the stateless forward -- same trajectory, and the reference's."""
    from oracle import draws, weights
    from oracle.make_golden import SEED, small_batch
    dev = _dev()
    over = dict(ew_net_type='r', x2h_out_fc=True)
    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    cfg.update(over)
    model = _model(weights.make_state_dict(SEED, cfg), **over)
    gs = load_golden('sample_ew_r_out_fc_4.npz')
    b = small_batch()[0].to(dev)
    outs = []
    for kw in (dict(use_graph=True), dict(use_graph=False), dict(use_session=False)):          # the captured step, launch by launch, stateless
        r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                                   torch.from_numpy(gs['init_ligand_pos']).to(dev), torch.from_numpy(gs['init_ligand_v']).to(dev),
                                   b.ligand_element_batch, num_steps=int(gs['steps']), center_pos_mode='protein',
                                   noise_source=draws.Source(int(gs['draws_base']), dev), **kw)
        assert np.array_equal(torch.stack(r['v_traj']).numpy(), gs['v_traj'].astype(np.int64)), kw
        close(torch.stack(r['pos_traj']), gs['pos_traj'], TOL_TRAJ, kw)
        outs.append(r)
    assert torch.equal(torch.stack(outs[0]['pos_traj']), torch.stack(outs[2]['pos_traj']))
    assert torch.equal(torch.stack(outs[0]['pos_traj']), torch.stack(outs[1]['pos_traj']))
    with pytest.raises(NotImplementedError, match='32 slots'):
        _model(weights.make_state_dict(SEED), ew_net_type='r', knn=48)
    # x2h_out_fc and sync_twoup are graph-agnostic: on a hybrid graph (chunk-walking kernels) the session and the stateless forward agree
    over2 = dict(x2h_out_fc=True, sync_twoup=True, cutoff_mode='hybrid')
    cfg2 = dict(weights.DEFAULT_MODEL_CONFIG)
    cfg2.update(over2)
    hyb = _model(weights.make_state_dict(SEED, cfg2), **over2)
    rs = [hyb.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                               torch.from_numpy(gs['init_ligand_pos']).to(dev), torch.from_numpy(gs['init_ligand_v']).to(dev),
                               b.ligand_element_batch, num_steps=3, center_pos_mode='protein',
                               noise_source=draws.Source(4700, dev), use_session=us) for us in (True, False)]
    assert torch.equal(torch.stack(rs[0]['pos_traj']), torch.stack(rs[1]['pos_traj']))
    assert torch.isfinite(torch.stack(rs[0]['pos_traj'])).all()
