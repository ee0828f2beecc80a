"""Oracle restatement vs the REAL reference model files on fresh inputs (beyond the committed golden vectors).

Runs only where the reference tree exists (the build container); skipped on the GPU box, where /root/reference is
absent by contract.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import reference_loader, weights
from oracle import restatement as R
from targetdiff_amd import workloads

pytestmark = pytest.mark.skipif(not reference_loader.available(), reason='reference tree not present')


def _maxdiff(a, b):
    return float((a.double() - b.double()).abs().max())


@pytest.fixture(scope='module')
def ref_model():
    from oracle import make_golden
    ref = reference_loader.load()
    model, sd = make_golden.build_reference_model(ref, seed=77)
    return ref, model, sd


def _batch(seed):
    pockets = [workloads.synthetic_pocket(500 + seed, 70, 3.0, 9.0), workloads.synthetic_pocket(600 + seed, 40, 3.0, 8.0)]
    b = workloads.pack_samples(pockets, 2, [8, 11, 6, 9])
    lpos, lv = workloads.init_ligand(b, generator=torch.Generator().manual_seed(seed), spread=1.5)
    return b, lpos, lv


@pytest.mark.parametrize('seed', [1, 2])
def test_forward_fresh_inputs(ref_model, seed):
    ref, model, sd = ref_model
    b, lpos, lv = _batch(seed)
    ppos, lposc, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    with torch.no_grad():
        want = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch)
    got = R.model_forward(sd, None, ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv,
                          b.ligand_element_batch)
    assert _maxdiff(got['pred_ligand_pos'], want['pred_ligand_pos']) < 2e-5
    assert _maxdiff(got['pred_ligand_v'], want['pred_ligand_v']) < 2e-5
    assert _maxdiff(got['final_h'], want['final_h']) < 2e-5


def test_likelihood_fresh_inputs(ref_model):
    ref, model, sd = ref_model
    b, lpos, lv = _batch(3)
    rec = {}
    o_normal, o_rand = torch.Tensor.normal_, torch.rand_like

    def normal_(self, *a, **k):
        out = o_normal(self, *a, **k)
        rec['noise'] = out.clone()
        return out

    def rand_like(x, *a, **k):
        out = o_rand(x, *a, **k)
        rec['uniform'] = out.clone()
        return out
    ts = torch.tensor([12, 0, 999, 400], dtype=torch.long)
    torch.Tensor.normal_, torch.rand_like = normal_, rand_like
    try:
        torch.manual_seed(5)
        want = model.likelihood_estimation(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                                           b.ligand_element_batch, ts)
    finally:
        torch.Tensor.normal_, torch.rand_like = o_normal, o_rand
    got = R.likelihood_estimation(sd, None, b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                                  b.ligand_element_batch, ts, noise=rec['noise'], uniform=rec['uniform'])
    for g, w in zip(got, want):
        np.testing.assert_allclose(g.numpy(), w.numpy(), rtol=1e-4, atol=1e-6)


def test_egnn_fresh_inputs(ref_model):
    import importlib
    ref, _, _ = ref_model
    egnn_mod = importlib.import_module('models.egnn')
    net = egnn_mod.EGNN(num_layers=4, hidden_dim=128, edge_feat_dim=4, num_r_gaussian=1, k=32, cutoff_mode='knn').eval()
    esd = weights.make_egnn_state_dict(9, num_layers=4)
    net.load_state_dict(esd, strict=True)
    b, lpos, _ = _batch(4)
    g = torch.Generator().manual_seed(10)
    h, x, batch, mask = R.compose_context(torch.randn(b.protein_pos.shape[0], 128, generator=g),
                                          torch.randn(lpos.shape[0], 128, generator=g), b.protein_pos, lpos,
                                          b.protein_element_batch, b.ligand_element_batch)
    with torch.no_grad():
        want = net(h, x, mask, batch)
    got = R.egnn_forward(esd, h, x, mask, batch, num_layers=4)
    assert _maxdiff(got['x'], want['x']) < 2e-5 and _maxdiff(got['h'], want['h']) < 5e-5


def test_mirror_loads_the_full_reference_state_dict_strict():
    """All 384 entries of the real reference model's state_dict (learnable tensors, the 15 frozen schedule Parameters,
    Lt_history / Lt_count, the unused init_h_emb_layer) load into the mirror with strict=True, as
    scripts/sample_diffusion.py:163 loads a checkpoint -- and the schedule constants the mirror builds itself are the
    reference's, bit for bit."""
    from oracle import reference_loader, shims, weights
    from targetdiff_amd.models import ScorePosNet3D
    ref = reference_loader.load()
    torch.manual_seed(123)
    ref_model = ref.ScorePosNet3D(shims.EasyDict(weights.DEFAULT_MODEL_CONFIG), weights.PROTEIN_FEATURE_DIM,
                                  weights.LIGAND_FEATURE_DIM)
    ref_sd = ref_model.state_dict()
    assert len(ref_sd) == 384
    mirror = ScorePosNet3D(dict(weights.DEFAULT_MODEL_CONFIG), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    own = {k: v.clone() for k, v in mirror.state_dict().items()}
    assert list(own) == list(ref_sd)                                    # same keys, same order
    frozen = [k for k, p in ref_model.named_parameters() if not p.requires_grad]
    assert len(frozen) == 15
    for k in frozen:
        assert torch.equal(own[k], ref_sd[k]), k
    res = mirror.load_state_dict(ref_sd, strict=True)
    assert not res.missing_keys and not res.unexpected_keys
    for k, v in mirror.state_dict().items():
        assert torch.equal(v, ref_sd[k]), k


def test_call_signatures_are_drop_in():
    """INTEGRATION.md route A is an import swap: every parameter of the reference's call signatures on this path exists in
    the mirror under the same name, in the same position and with the same default (the mirror may add keyword-only
    extras at the end)."""
    import inspect
    from oracle import reference_loader
    from targetdiff_amd import models as M
    from targetdiff_amd import sampling as S
    ref = reference_loader.load()
    drv = reference_loader.load_driver()
    pairs = [(ref.ScorePosNet3D.__init__, M.ScorePosNet3D.__init__), (ref.ScorePosNet3D.forward, M.ScorePosNet3D.forward),
             (ref.ScorePosNet3D.sample_diffusion, M.ScorePosNet3D.sample_diffusion),
             (ref.ScorePosNet3D.fetch_embedding, M.ScorePosNet3D.fetch_embedding),
             (ref.ScorePosNet3D.likelihood_estimation, M.ScorePosNet3D.likelihood_estimation),
             (ref.get_refine_net, M.get_refine_net), (drv.sample_diffusion_ligand, S.sample_diffusion_ligand),
             (drv.unbatch_v_traj, S.unbatch_v_traj)]
    for f_ref, f_own in pairs:
        want = list(inspect.signature(inspect.unwrap(f_ref)).parameters.values())
        got = list(inspect.signature(inspect.unwrap(f_own)).parameters.values())
        assert len(got) >= len(want), f_ref.__qualname__
        for a, b in zip(want, got):
            assert a.name == b.name and a.default == b.default, (f_ref.__qualname__, a, b)
        for extra in got[len(want):]:
            assert extra.default is not inspect.Parameter.empty, (f_own.__qualname__, extra)      # optional additions only
    # the refine net seam (models/uni_transformer.py:301) and the attributes the drivers read
    import importlib
    ut = importlib.import_module('models.uni_transformer')
    want = [p.name for p in inspect.signature(ut.UniTransformerO2TwoUpdateGeneral.forward).parameters.values()]
    got = [p.name for p in inspect.signature(M.UniTransformerO2TwoUpdateGeneral.forward).parameters.values()]
    assert got[:len(want)] == want
    want = list(inspect.signature(ut.UniTransformerO2TwoUpdateGeneral.__init__).parameters.values())
    got = list(inspect.signature(M.UniTransformerO2TwoUpdateGeneral.__init__).parameters.values())
    for a, b in zip(want, got):
        assert a.name == b.name and a.default == b.default, (a, b)
    from oracle import weights
    m = M.ScorePosNet3D(dict(weights.DEFAULT_MODEL_CONFIG), 27, 13)
    assert m.num_classes == 13 and m.num_timesteps == 1000 and m.center_pos_mode == 'protein'


def test_unsorted_batch_vectors_forward_and_sampling_loop(ref_model):
    """Unsorted batch vectors are legal input of the reference (compose_context stable-sorts, models/common.py:126): the forward
    answers in compose_context's order (the ligand rows are NOT put back in input order), and the sampling loop combines those
    answers element by element with its state in input order.  The restatement follows both, which is what the GPU tests then
    hold the HIP path to (tests/test_gpu_long_parity.py::test_unsorted_batch_vectors_*)."""
    ref, model, sd = ref_model
    b, lpos, lv = _batch(4)
    g = torch.Generator().manual_seed(12)
    pp, pl = torch.randperm(b.protein_pos.shape[0], generator=g), torch.randperm(lpos.shape[0], generator=g)
    ppos, pv, bp = b.protein_pos[pp], b.protein_atom_feature.float()[pp], b.protein_element_batch[pp]
    xl, vl, bl = lpos[pl], lv[pl], b.ligand_element_batch[pl]
    cp, cl, off = ref.center_pos(ppos, xl, bp, bl, mode='protein')
    with torch.no_grad():
        want = model(cp, pv, bp, cl, vl, bl)
        sp, sl = torch.sort(bp, stable=True).indices, torch.sort(bl, stable=True).indices
        pre = model(cp[sp], pv[sp], bp[sp], cl[sl], vl[sl], bl[sl])
    assert torch.equal(want['pred_ligand_pos'], pre['pred_ligand_pos']) and torch.equal(want['final_h'], pre['final_h'])
    got = R.model_forward(sd, None, cp, pv, bp, cl, vl, bl)
    assert _maxdiff(got['pred_ligand_pos'], want['pred_ligand_pos']) < 2e-5
    assert _maxdiff(got['pred_ligand_v'], want['pred_ligand_v']) < 2e-5
    # three steps of the reference's own loop on the unsorted vectors vs the restatement's loop on the same draws
    from oracle import draws
    from oracle.make_golden_r2 import counter_draws
    steps = 3
    with counter_draws(9100), torch.no_grad():
        r = model.sample_diffusion(ppos, pv, bp, xl, vl, bl, num_steps=steps, center_pos_mode='protein')
    sched = R.diffusion_schedules(dict(weights.DEFAULT_MODEL_CONFIG))
    rcp, rcl, roff = R.center_positions(ppos, xl, bp, bl)
    x, v = rcl.clone(), vl.clone()
    for s, t in enumerate(reversed(range(1000 - steps, 1000))):
        preds = R.model_forward(sd, None, rcp, pv, bp, x, v, bl)                 # answers in compose_context's order ...
        tt = torch.full((int(bp.max()) + 1,), t, dtype=torch.long)
        x, v, _, _ = R.posterior_step(sched, tt, x, v, preds['pred_ligand_pos'], preds['pred_ligand_v'], bl,    # ... met by the state in input order
                                      draws.normal(9100, s, tuple(x.shape)), draws.uniform(9101, s, (x.shape[0], 13)), 13)
        assert _maxdiff(r['pos_traj'][s], x + roff[bl]) < 5e-5, s
        assert torch.equal(r['v_traj'][s], v), s
