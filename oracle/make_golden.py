"""Generate tests/golden/*.npz by running the REAL reference model files (TEST INFRASTRUCTURE).

Run in the build container only (needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden

The reference ships no golden vectors, known-answer tests or checkpoints for this path (SURVEY.md
section 4 / 8c), so these fixtures are outputs of the reference's own
``models/{molopt_score_model,uni_transformer,common}.py`` imported unmodified (third-party ops shimmed
by ``oracle/shims.py``) on seeded weights (``oracle/weights.py``) and seeded inputs.  They pin
``oracle/restatement.py`` (tests/test_oracle_golden.py) and, through it, the HIP path.
"""
from __future__ import annotations

import os
import sys

import numpy as np
import torch

from . import reference_loader, shims, weights
from targetdiff_amd import workloads

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
SEED = 2021
KEEP_EXISTING = '--keep-existing' in sys.argv      # only write fixtures that are not in tests/golden yet


def _save(path, **arrays):
    """np.savez_compressed, except that --keep-existing leaves committed fixtures byte-identical (zip timestamps)."""
    if KEEP_EXISTING and os.path.exists(path):
        with np.load(path) as z:
            for k, v in arrays.items():
                assert np.array_equal(z[k], np.asarray(v)), (path, k, 'regenerated fixture differs from the committed one')
        print('kept', os.path.basename(path), '(regenerated values identical)')
        return
    np.savez_compressed(path, **arrays)


def build_reference_model(ref, seed=SEED):
    cfg = shims.EasyDict(weights.DEFAULT_MODEL_CONFIG)
    model = ref.ScorePosNet3D(cfg, weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    sd = weights.make_state_dict(seed)
    ref_sd = model.state_dict()
    learnable = {k for k, p in model.named_parameters() if p.requires_grad}
    learnable |= {k for k in ref_sd if k.endswith('distance_expansion.offset')}
    assert set(sd) == learnable, (set(sd) ^ learnable)
    for k in sd:
        assert tuple(sd[k].shape) == tuple(ref_sd[k].shape), k
        if k.endswith('offset'):
            assert torch.equal(sd[k], ref_sd[k]), k
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys
    return model.eval(), sd


def small_batch(seed=7):
    """Three graphs incl. one with fewer than k+1 = 33 nodes (fewer than k in-edges per node)."""
    pockets = [workloads.synthetic_pocket(101, 60, 3.0, 9.0), workloads.synthetic_pocket(102, 45, 3.0, 8.0),
               workloads.synthetic_pocket(103, 20, 2.5, 6.0)]
    sizes = [9, 7, 6]
    b = workloads.pack_samples(pockets, 1, sizes)
    g = torch.Generator().manual_seed(seed)
    pos, v = workloads.init_ligand(b, generator=g)
    return b, pos, v


def ref_forward_with_intermediates(ref, model, b, lpos, lv, centre=True):
    ppos = b.protein_pos
    if centre:
        ppos, lpos, _ = ref.center_pos(ppos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    pv = b.protein_atom_feature.float()
    # intermediates through forward hooks on the real modules
    inter = {'h_layers': [], 'x_layers': []}
    hooks = []
    def layer_hook(m, i, o):            # must return None (a returned value would replace the output)
        inter['h_layers'].append(o[0].detach().clone())
        inter['x_layers'].append(o[1].detach().clone())

    def gate_hook(m, i, o):
        inter['e_w_logits'] = o.detach().clone()
    for layer in model.refine_net.base_block:
        hooks.append(layer.register_forward_hook(layer_hook))
    hooks.append(model.refine_net.edge_pred_layer.register_forward_hook(gate_hook))
    orig_connect = model.refine_net._connect_edge

    def connect(x, mask_ligand, batch):
        ei = orig_connect(x, mask_ligand, batch)
        inter['edge_index'] = ei.clone()
        inter['mask_ligand'] = mask_ligand.clone()
        inter['batch_all'] = batch.clone()
        return ei
    model.refine_net._connect_edge = connect
    with torch.no_grad():
        preds = model(ppos, pv, b.protein_element_batch, lpos, lv, b.ligand_element_batch,
                      time_step=torch.zeros(b.num_graphs, dtype=torch.long))
    model.refine_net._connect_edge = orig_connect
    for hk in hooks:
        hk.remove()
    return ppos, lpos, preds, inter


def edge_index_to_table(edge_index, N, k):
    src, dst = edge_index
    nbr = torch.full((N, k), -1, dtype=torch.long)
    slot = torch.zeros(N, dtype=torch.long)
    for s, d in zip(src.tolist(), dst.tolist()):
        nbr[d, slot[d]] = s
        slot[d] += 1
    return nbr


def main():
    os.makedirs(GOLDEN_DIR, exist_ok=True)
    ref = reference_loader.load()
    torch.set_num_threads(8)
    model, sd = build_reference_model(ref)
    k = weights.DEFAULT_MODEL_CONFIG['knn']

    # ---------------------------------------------------------------- schedules (molopt_score_model.py:221-267)
    names = ['betas', 'alphas_cumprod', 'posterior_mean_c0_coef', 'posterior_mean_ct_coef', 'posterior_logvar',
             'log_alphas_v', 'log_one_minus_alphas_v', 'log_alphas_cumprod_v', 'log_one_minus_alphas_cumprod_v']
    _save(os.path.join(GOLDEN_DIR, 'schedules.npz'),
                        **{n: getattr(model, n).detach().numpy() for n in names})

    # ---------------------------------------------------------------- 1h36 pocket (examples/, real geometry)
    pdb = os.path.join(reference_loader.REFERENCE_ROOT, 'examples',
                       '1h36_A_rec_1h36_r88_lig_tt_docked_0_pocket10.pdb')
    pocket = workloads.pocket_from_pdb(pdb, '1h36_pocket10')
    sys.path.insert(0, reference_loader.REFERENCE_ROOT)
    from utils.evaluation import atom_num                         # reference ligand-size prior
    space = atom_num.get_space_size(pocket.pos)
    np.random.seed(SEED)
    sizes100 = np.array([atom_num.sample_atom_num(space).astype(int) for _ in range(100)])
    _save(os.path.join(GOLDEN_DIR, 'pocket_1h36.npz'), pos=pocket.pos,
                        feat=pocket.feat.astype(np.int8), prior_sizes_seed2021=sizes100, space_size=space)
    print('1h36:', pocket.num_atoms, 'atoms; space size', space, 'sizes[:8]', sizes100[:8], 'mean', sizes100.mean())

    # ---------------------------------------------------------------- forward, small synthetic (all stages)
    b, lpos, lv = small_batch()
    ppos, lpos_c, preds, inter = ref_forward_with_intermediates(ref, model, b, lpos, lv)
    N = ppos.shape[0] + lpos_c.shape[0]
    nbr = edge_index_to_table(inter['edge_index'], N, k)
    e_w = torch.sigmoid(inter['e_w_logits']).squeeze(-1)
    ew_tab = torch.zeros(N, k)
    dst = inter['edge_index'][1]
    slot = torch.zeros(N, dtype=torch.long)
    for e, d in enumerate(dst.tolist()):
        ew_tab[d, slot[d]] = e_w[e]
        slot[d] += 1
    _save(
        os.path.join(GOLDEN_DIR, 'forward_small.npz'),
        protein_pos=ppos.numpy(), protein_feat=b.protein_atom_feature.numpy().astype(np.int8),
        batch_protein=b.protein_element_batch.numpy(), ligand_pos=lpos_c.numpy(), ligand_v=lv.numpy(),
        batch_ligand=b.ligand_element_batch.numpy(),
        nbr=nbr.numpy().astype(np.int32), e_w=ew_tab.numpy(), mask_ligand=inter['mask_ligand'].numpy(),
        h_layers=torch.stack(inter['h_layers']).numpy(), x_layers=torch.stack(inter['x_layers']).numpy(),
        pred_ligand_pos=preds['pred_ligand_pos'].numpy(), pred_ligand_v=preds['pred_ligand_v'].numpy(),
        final_h=preds['final_h'].numpy(), final_ligand_h=preds['final_ligand_h'].numpy())
    print('forward_small: N =', N, 'edges =', inter['edge_index'].shape[1])

    # fix_x=True variant (fetch_embedding path, molopt_score_model.py:620-631)
    with torch.no_grad():
        pe = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lpos_c, lv,
                   b.ligand_element_batch, fix_x=True)
    _save(os.path.join(GOLDEN_DIR, 'forward_small_fixx.npz'),
                        pred_ligand_pos=pe['pred_ligand_pos'].numpy(), pred_ligand_v=pe['pred_ligand_v'].numpy(),
                        final_ligand_h=pe['final_ligand_h'].numpy())

    # ---------------------------------------------------------------- forward, 1h36 x 2 samples (outputs only)
    b2 = workloads.pack_samples(pocket, 2, sizes100[:2])
    g = torch.Generator().manual_seed(11)
    lpos2, lv2 = workloads.init_ligand(b2, generator=g)
    ppos2, lpos2c, preds2, inter2 = ref_forward_with_intermediates(ref, model, b2, lpos2, lv2)
    N2 = ppos2.shape[0] + lpos2c.shape[0]
    nbr2 = edge_index_to_table(inter2['edge_index'], N2, k)
    _save(
        os.path.join(GOLDEN_DIR, 'forward_1h36x2.npz'),
        ligand_pos=lpos2c.numpy(), ligand_v=lv2.numpy(), sizes=sizes100[:2],
        nbr=nbr2.numpy().astype(np.int32),
        pred_ligand_pos=preds2['pred_ligand_pos'].numpy(), pred_ligand_v=preds2['pred_ligand_v'].numpy(),
        final_ligand_h=preds2['final_ligand_h'].numpy(),
        final_h_sample=preds2['final_h'][::16].numpy())
    print('forward_1h36x2: N =', N2)

    # ---------------------------------------------------------------- sample_diffusion, 6 steps, recorded RNG
    rec = {'randn': [], 'rand': []}
    o_randn, o_rand = torch.randn_like, torch.rand_like

    def randn_like(x, *a, **kw):
        out = o_randn(x, *a, **kw)
        rec['randn'].append(out.clone())
        return out

    def rand_like(x, *a, **kw):
        out = o_rand(x, *a, **kw)
        rec['rand'].append(out.clone())
        return out
    torch.randn_like, torch.rand_like = randn_like, rand_like
    try:
        torch.manual_seed(SEED)
        with torch.no_grad():
            r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                                       lpos, lv, b.ligand_element_batch, num_steps=6, center_pos_mode='protein')
    finally:
        torch.randn_like, torch.rand_like = o_randn, o_rand
    _save(
        os.path.join(GOLDEN_DIR, 'sample_small.npz'),
        init_ligand_pos=lpos.numpy(), init_ligand_v=lv.numpy(),
        noises=torch.stack(rec['randn']).numpy(), uniforms=torch.stack(rec['rand']).numpy(),
        pos=r['pos'].numpy(), v=r['v'].numpy(), pos_traj=torch.stack(r['pos_traj']).numpy(),
        v_traj=torch.stack(r['v_traj']).numpy(), v0_traj=torch.stack(r['v0_traj']).numpy(),
        vt_traj=torch.stack(r['vt_traj']).numpy())
    print('sample_small: steps', len(rec['randn']))

    # ---------------------------------------------------------------- posterior known-answer test, mixed t
    # (incl. the noiseless t == 0 branch); the reference loop body molopt_score_model.py:673-685 called
    # through the model's own methods.
    gk = torch.Generator().manual_seed(5)
    bl = b.ligand_element_batch
    n_l = bl.numel()
    t = torch.tensor([0, 1, 537], dtype=torch.long)
    x_t = torch.randn(n_l, 3, generator=gk)
    x0 = x_t + 0.3 * torch.randn(n_l, 3, generator=gk)
    v_t = torch.randint(0, 13, (n_l,), generator=gk)
    v0_logits = 2.0 * torch.randn(n_l, 13, generator=gk)
    noise = torch.randn(n_l, 3, generator=gk)
    uni = torch.rand(n_l, 13, generator=gk)
    with torch.no_grad():
        mean = model.q_pos_posterior(x0=x0, xt=x_t, t=t, batch=bl)
        logvar = ref.extract(model.posterior_logvar, t, bl)
        nz = (1 - (t == 0).float())[bl].unsqueeze(-1)
        pos_next = mean + nz * (0.5 * logvar).exp() * noise
        log_v0 = torch.log_softmax(v0_logits, dim=-1)
        log_vt = ref.index_to_log_onehot(v_t, 13)
        log_post = model.q_v_posterior(log_v0, log_vt, t, bl)
        gumbel = -torch.log(-torch.log(uni + 1e-30) + 1e-30)
        v_next = (gumbel + log_post).argmax(dim=-1)
    _save(os.path.join(GOLDEN_DIR, 'posterior_kat.npz'), t=t.numpy(), batch_ligand=bl.numpy(),
                        x_t=x_t.numpy(), x0=x0.numpy(), v_t=v_t.numpy(), v0_logits=v0_logits.numpy(),
                        noise=noise.numpy(), uniform=uni.numpy(), pos_next=pos_next.numpy(),
                        v_next=v_next.numpy(), log_v0=log_v0.numpy(), log_post=log_post.numpy())
    # ---------------------------------------------------------------- return_all (num_blocks = 1: block input + output)
    ppos_c, lpos_c, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    with torch.no_grad():
        pa = model(ppos_c, b.protein_atom_feature.float(), b.protein_element_batch, lpos_c, lv,
                   b.ligand_element_batch, return_all=True)
    _save(os.path.join(GOLDEN_DIR, 'forward_small_return_all.npz'),
                        layer_pred_ligand_pos=torch.stack(pa['layer_pred_ligand_pos']).numpy(),
                        layer_pred_ligand_v=torch.stack(pa['layer_pred_ligand_v']).numpy())
    print('return_all: layers', len(pa['layer_pred_ligand_pos']))

    # ---------------------------------------------------------------- likelihood_estimation, recorded RNG
    # (scripts/likelihood_est_diffusion.py:30,48): mixed time steps incl. the t == 0 decoder branch, and the prior
    rec = {'normal': [], 'rand': []}
    o_normal, o_rand = torch.Tensor.normal_, torch.rand_like

    def normal_(self, *a, **kw):
        out = o_normal(self, *a, **kw)
        rec['normal'].append(out.clone())
        return out

    def rand_like2(x, *a, **kw):
        out = o_rand(x, *a, **kw)
        rec['rand'].append(out.clone())
        return out
    torch.Tensor.normal_, torch.rand_like = normal_, rand_like2
    try:
        torch.manual_seed(SEED + 1)
        tl = torch.tensor([0, 1, 537], dtype=torch.long)
        kl_pos, kl_v = model.likelihood_estimation(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch,
                                                   lpos, lv, b.ligand_element_batch, tl)
        tT = torch.full((3,), 1000, dtype=torch.long)
        klp_prior, klv_prior = model.likelihood_estimation(b.protein_pos, b.protein_atom_feature.float(),
                                                           b.protein_element_batch, lpos, lv, b.ligand_element_batch, tT)
    finally:
        torch.Tensor.normal_, torch.rand_like = o_normal, o_rand
    assert len(rec['normal']) == 1 and len(rec['rand']) == 1
    _save(os.path.join(GOLDEN_DIR, 'likelihood_small.npz'), time_step=tl.numpy(),
                        protein_pos=b.protein_pos.numpy(), protein_feat=b.protein_atom_feature.numpy().astype(np.int8),
                        batch_protein=b.protein_element_batch.numpy(), batch_ligand=b.ligand_element_batch.numpy(),
                        ligand_pos=lpos.numpy(), ligand_v=lv.numpy(), noise=rec['normal'][0].numpy(),
                        uniform=rec['rand'][0].numpy(), kl_pos=kl_pos.numpy(), kl_v=kl_v.numpy(),
                        kl_pos_prior=klp_prior.numpy(), kl_v_prior=klv_prior.numpy())
    print('likelihood_small:', kl_pos.numpy(), kl_v.numpy(), klp_prior.numpy(), klv_prior.numpy())

    # ---------------------------------------------------------------- standalone EGNN refine net (models/egnn.py)
    # built exactly as get_refine_net('egnn', config) does (models/molopt_score_model.py:34-42); seeded features on the
    # small batch's composed geometry
    import importlib
    egnn_mod = importlib.import_module('models.egnn')
    L_e = weights.DEFAULT_MODEL_CONFIG['num_layers']
    enet = egnn_mod.EGNN(num_layers=L_e, hidden_dim=128, edge_feat_dim=4, num_r_gaussian=1, k=k, cutoff_mode='knn').eval()
    esd = weights.make_egnn_state_dict(SEED, num_layers=L_e)
    assert set(esd) == set(enet.state_dict()), set(esd) ^ set(enet.state_dict())
    enet.load_state_dict(esd, strict=True)
    from . import restatement as Rst
    ppos_e, lpos_e, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    ge = torch.Generator().manual_seed(SEED + 2)
    n_p, n_l = ppos_e.shape[0], lpos_e.shape[0]
    h_ctx, x_ctx, batch_ctx, mask_ctx = Rst.compose_context(torch.randn(n_p, 128, generator=ge), torch.randn(n_l, 128, generator=ge),
                                                            ppos_e, lpos_e, b.protein_element_batch, b.ligand_element_batch)
    with torch.no_grad():
        oe = enet(h_ctx, x_ctx, mask_ctx, batch_ctx, return_all=True)
    _save(os.path.join(GOLDEN_DIR, 'egnn_small.npz'), h=h_ctx.numpy(), x=x_ctx.numpy(), batch=batch_ctx.numpy(),
          mask_ligand=mask_ctx.numpy(), num_layers=np.int64(L_e), all_x=torch.stack(oe['all_x']).numpy(),
          h_layer1=oe['all_h'][1].numpy(), h_layer5=oe['all_h'][5].numpy(), h_final=oe['h'].numpy())
    print('egnn_small: N =', h_ctx.shape[0], 'max |dx| =', float((oe['x'] - x_ctx).abs().max()))

    for f in sorted(os.listdir(GOLDEN_DIR)):
        print(f, os.path.getsize(os.path.join(GOLDEN_DIR, f)) // 1024, 'KiB')


if __name__ == '__main__':
    main()
