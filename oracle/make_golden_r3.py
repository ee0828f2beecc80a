"""Round-3 fixtures (TEST INFRASTRUCTURE; build container only, needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_r3 [name ...] [--keep-existing]

The C5 sweep at C5 size and late-t geometry on the real pocket:

  forward_c5_k48, forward_c5_k64, forward_c5_hybrid
                      one denoiser forward of the REAL reference on the C5-shaped pack of forward_c5.npz (1000-atom pocket
                      x 2, ligands of 150 and 30 atoms, same ligand state) with knn = 48 / 64 and cutoff_mode = 'hybrid'
                      (models/uni_transformer.py:276-286, models/common.py:165-212): 2 chunks per row for the k-NN graphs,
                      6 chunks on the 150-atom ligand's hybrid rows, multi-pass knn_general_kernel (> 704 nodes per graph).
  forward_c5_radius   the same pack on a radius graph r = 6 A, fan-out cap 48 -- the reference's radius mode is dead code
                      (`self.r` unassigned, models/uni_transformer.py:278), so this one comes from oracle/restatement.py
                      under the project's rule (oracle/shims.py radius_neighbours) and says so in the file (`source`).
  sample_1h36x2_hybrid_20
                      20 reverse steps (t = 999 .. 980) of the reference's own loop on 1h36 x 2 with cutoff_mode = 'hybrid'
                      and the counter draws: sampling on a general graph against the reference, not the restatement.
  forward_other_weights
                      forward_small_seed7 / seed11: the small batch under two other seeded weight sets (gain 1.8 / 0.5).
  sample_1h36x2_1000  a complete 1000-step run of the reference on the real pocket (1h36 x 2, prior sizes, k = 32) with the
                      counter draws: late-t arithmetic at real-pocket density against the session caching.  Every step's
                      positions / types, the log-probabilities of every 50th step and of the last 12.
"""
from __future__ import annotations

import os
import sys
import time

import numpy as np
import torch

from . import reference_loader, shims, weights
from .make_golden import GOLDEN_DIR, SEED, _save, build_reference_model, ref_forward_with_intermediates
from .make_golden_r2 import C5_POCKET, C5_SIZES, counter_draws, edges_to_csr, load_1h36
from targetdiff_amd import workloads


def c5_pack():
    """The pack and ligand state of forward_c5.npz (oracle/make_golden_r2.py gen_forward_c5)."""
    pocket = workloads.synthetic_pocket(**C5_POCKET)
    b = workloads.pack_samples(pocket, 2, C5_SIZES)
    g = torch.Generator().manual_seed(SEED + 12)
    lpos, lv = workloads.init_ligand(b, generator=g, spread=3.0)
    return b, lpos, lv


def variant_model(ref, cfg_update):
    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    cfg.update(cfg_update)
    model = ref.ScorePosNet3D(shims.EasyDict(cfg), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    res = model.load_state_dict(weights.make_state_dict(SEED), strict=False)
    assert not res.unexpected_keys
    return model.eval()


def gen_c5_variant(ref, name, cfg_update):
    b, lpos, lv = c5_pack()
    model = variant_model(ref, cfg_update)
    t0 = time.time()
    ppos, lpos_c, preds, inter = ref_forward_with_intermediates(ref, model, b, lpos, lv)
    N = ppos.shape[0] + lpos_c.shape[0]
    row_ptr, col = edges_to_csr(inter['edge_index'], N)
    _save(os.path.join(GOLDEN_DIR, name + '.npz'), source=np.asarray('reference'), row_ptr=row_ptr, col=col.astype(np.int16),
          ligand_pos=lpos_c.numpy(), ligand_v=lv.numpy().astype(np.int8), protein_pos_centred=ppos.numpy(),
          pred_ligand_pos=preds['pred_ligand_pos'].numpy(), pred_ligand_v=preds['pred_ligand_v'].numpy(),
          final_ligand_h=preds['final_ligand_h'].numpy(), final_h_sample=preds['final_h'][::16].numpy(),
          x_last=inter['x_layers'][-1].numpy())
    deg = np.diff(row_ptr)
    print(f'{name}: N = {N}, E = {len(col)}, degree min/max = {deg.min()}/{deg.max()}, {time.time() - t0:.1f} s')


def gen_forward_c5_k48(ref, model):
    gen_c5_variant(ref, 'forward_c5_k48', dict(knn=48))


def gen_forward_c5_k64(ref, model):
    gen_c5_variant(ref, 'forward_c5_k64', dict(knn=64))


def gen_forward_c5_hybrid(ref, model):
    gen_c5_variant(ref, 'forward_c5_hybrid', dict(cutoff_mode='hybrid'))


C5_RADIUS = dict(cutoff_mode='radius', r=6.0, max_num_neighbors=48)


def gen_forward_c5_radius(ref, model):
    from . import restatement as R
    b, lpos, lv = c5_pack()
    sd = weights.make_state_dict(SEED)
    cfg = dict(weights.DEFAULT_MODEL_CONFIG, **C5_RADIUS)
    ppos, lpos_c, _ = R.center_positions(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch)
    t0 = time.time()
    want = R.model_forward(sd, cfg, ppos, b.protein_atom_feature.float(), b.protein_element_batch, lpos_c, lv, b.ligand_element_batch)
    # the neighbour table of the project's radius rule on the composed coordinates
    x = torch.cat([ppos[:1000], lpos_c[:C5_SIZES[0]], ppos[1000:], lpos_c[C5_SIZES[0]:]])
    batch = torch.repeat_interleave(torch.arange(2), torch.tensor([1000 + C5_SIZES[0], 1000 + C5_SIZES[1]]))
    table = shims.radius_neighbours(x, C5_RADIUS['r'], batch, C5_RADIUS['max_num_neighbors'])
    _save(os.path.join(GOLDEN_DIR, 'forward_c5_radius.npz'), source=np.asarray('oracle/restatement.py (project rule)'),
          table=table.numpy().astype(np.int16), ligand_pos=lpos_c.numpy(), ligand_v=lv.numpy().astype(np.int8),
          protein_pos_centred=ppos.numpy(), pred_ligand_pos=want['pred_ligand_pos'].numpy(),
          pred_ligand_v=want['pred_ligand_v'].numpy(), final_ligand_h=want['final_ligand_h'].numpy(),
          final_h_sample=want['final_h'][::16].numpy())
    deg = (table >= 0).sum(1)
    print(f'forward_c5_radius: degree min/mean/max = {int(deg.min())}/{float(deg.float().mean()):.1f}/{int(deg.max())}, {time.time() - t0:.1f} s')


def _pack_1h36x2():
    pocket, sizes = load_1h36()
    b = workloads.pack_samples(pocket, 2, sizes[:2])
    return b, sizes[:2]


def _run_reference(model, b, lpos, lv, steps, base):
    with counter_draws(base), torch.no_grad():
        return model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                                      b.ligand_element_batch, num_steps=steps, center_pos_mode='protein')


def gen_sample_1h36x2_hybrid_20(ref, model):
    b, sizes = _pack_1h36x2()
    g = torch.Generator().manual_seed(SEED + 31)
    lpos, lv = workloads.init_ligand(b, generator=g)
    m = variant_model(ref, dict(cutoff_mode='hybrid'))
    t0 = time.time()
    r = _run_reference(m, b, lpos, lv, 20, 6100)
    keep = [0, 1, 10, 19]
    _save(os.path.join(GOLDEN_DIR, 'sample_1h36x2_hybrid_20.npz'), sizes=np.asarray(sizes), draws_base=np.int64(6100),
          init_ligand_pos=lpos.numpy(), init_ligand_v=lv.numpy().astype(np.int8),
          pos_traj=torch.stack(r['pos_traj']).numpy(), v_traj=torch.stack(r['v_traj']).numpy().astype(np.int8),
          pos=r['pos'].numpy(), v=r['v'].numpy().astype(np.int8), kept_steps=np.asarray(keep),
          v0_traj=torch.stack([r['v0_traj'][s] for s in keep]).numpy(),
          vt_traj=torch.stack([r['vt_traj'][s] for s in keep]).numpy())
    print(f'sample_1h36x2_hybrid_20: N_l = {lpos.shape[0]}, 20 steps in {time.time() - t0:.0f} s')


def gen_sample_1h36x2_1000(ref, model):
    b, sizes = _pack_1h36x2()
    g = torch.Generator().manual_seed(SEED + 32)
    lpos, lv = workloads.init_ligand(b, generator=g)
    t0 = time.time()
    r = _run_reference(model, b, lpos, lv, 1000, 7100)
    keep = sorted(set(list(range(49, 1000, 50)) + list(range(988, 1000))))
    _save(os.path.join(GOLDEN_DIR, 'sample_1h36x2_1000.npz'), sizes=np.asarray(sizes), draws_base=np.int64(7100),
          init_ligand_pos=lpos.numpy(), init_ligand_v=lv.numpy().astype(np.int8),
          pos_traj=torch.stack(r['pos_traj']).numpy(), v_traj=torch.stack(r['v_traj']).numpy().astype(np.int8),
          pos=r['pos'].numpy(), v=r['v'].numpy().astype(np.int8), kept_steps=np.asarray(keep),
          v0_traj=torch.stack([r['v0_traj'][s] for s in keep]).numpy(),
          vt_traj=torch.stack([r['vt_traj'][s] for s in keep]).numpy())
    print(f'sample_1h36x2_1000: N_l = {lpos.shape[0]}, 1000 steps in {time.time() - t0:.0f} s')


OTHER_WEIGHTS = [(7, 1.8), (11, 0.5)]


def gen_forward_other_weights(ref, model):
    """The small 3-graph batch under other seeded weight sets of the REAL reference (another seed with a stronger
    non-linearity / larger coordinate updates, another with a weaker one): parity must not hang on the one weight set."""
    from .make_golden import small_batch
    b, lpos, lv = small_batch()
    for seed, gain in OTHER_WEIGHTS:
        m = ref.ScorePosNet3D(shims.EasyDict(dict(weights.DEFAULT_MODEL_CONFIG)), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
        res = m.load_state_dict(weights.make_state_dict(seed, gain=gain), strict=False)
        assert not res.unexpected_keys
        m.eval()
        ppos, lpos_c, preds, inter = ref_forward_with_intermediates(ref, m, b, lpos, lv)
        _save(os.path.join(GOLDEN_DIR, f'forward_small_seed{seed}.npz'), seed=np.int64(seed), gain=np.float64(gain),
              protein_pos_centred=ppos.numpy(), ligand_pos=lpos_c.numpy(), ligand_v=lv.numpy().astype(np.int8),
              pred_ligand_pos=preds['pred_ligand_pos'].numpy(), pred_ligand_v=preds['pred_ligand_v'].numpy(),
              final_h=preds['final_h'].numpy())
        print(f'forward_small_seed{seed}: gain {gain}, max |h| {float(preds["final_h"].abs().max()):.2f}')


GENERATORS = {'forward_c5_k48': gen_forward_c5_k48, 'forward_c5_k64': gen_forward_c5_k64,
              'forward_c5_hybrid': gen_forward_c5_hybrid, 'forward_c5_radius': gen_forward_c5_radius,
              'sample_1h36x2_hybrid_20': gen_sample_1h36x2_hybrid_20, 'sample_1h36x2_1000': gen_sample_1h36x2_1000,
              'forward_other_weights': gen_forward_other_weights}


def main():
    names = [a for a in sys.argv[1:] if not a.startswith('--')] or list(GENERATORS)
    ref = reference_loader.load()
    torch.set_num_threads(int(os.environ.get('TD_GOLDEN_THREADS', '8')))
    model, _ = build_reference_model(ref)
    for n in names:
        GENERATORS[n](ref, model)


if __name__ == '__main__':
    main()
