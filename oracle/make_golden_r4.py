"""Round-4 fixtures from the REAL reference (TEST INFRASTRUCTURE; build container only):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_r4 [--keep-existing]

Configurations of ScorePosNet3D outside configs/training.yml that the mirror accepts since round 4:
  forward_time_simple.npz    time_emb_dim > 0, time_emb_mode = 'simple' (models/molopt_score_model.py:286-303, 319-329): forward
      (return_all) on the small batch with a different time step per graph.  ('sin' cannot be run: the reference concatenates
      a [B, dim] feature with the [N_l, C] one-hot, :326-327, and raises -- checked here.)
  sample_time_simple_6.npz   6 reverse steps of the reference's loop with that embedding (it changes every step), counter draws;
  sample_noise_6.npz         6 reverse steps with model_mean_type = 'noise' (:663-666);
  forward_blocks2.npz / sample_blocks2_4.npz   num_blocks = 2 (models/uni_transformer.py:306-323: the nine layers applied twice, graph and
      edge gate rebuilt from the moved coordinates in between): forward on the small batch, 4 reverse steps.
  forward_ln_signs.npz       the default architecture with LayerNorm weights of every sign (oracle.weights.ln_signs_state_dict: negative,
      zero and tiny entries in every MLP): forward (return_all) on the small batch.  Pins the LayerNorm fold of the packed edge MLPs.
  forward_ew_r_out_fc.npz / forward_ew_none.npz / forward_out_fc.npz / forward_ew_m.npz / sample_ew_r_out_fc_4.npz   the gate and output options of the
      attention layers outside configs/training.yml: ew_net_type = 'r' (every stage's own Linear(80, 1) + sigmoid on the layer's radial
      features, models/uni_transformer.py:34-35, 60-61, 102-103, 124-125), 'm' (the x2h gate from the edge's value vector, :36-37, 62-63),
      anything else but 'global' (e_w = 1, :64-67), and
      x2h_out_fc = True (node_output([attention output | h]) + h, :39-40, 81-84).  'r' + out_fc are the reference CLASS's defaults.
  forward_sync_twoup.npz     sync_twoup = True: the h2x stage reads the layer's input features (:198).
  forward_stages_2_2.npz / forward_stages_1_3_r.npz   several stages per layer (num_x2h / num_h2x, :190-206).
Weights: oracle.weights.time_emb_state_dict / make_state_dict(seed, cfg) (seeded per key; the fixtures hold outputs only)."""
from __future__ import annotations

import os

import numpy as np
import torch

from . import reference_loader, shims, weights
from .make_golden import GOLDEN_DIR, SEED, _save, small_batch
from .make_golden_r2 import counter_draws

TIME_STEPS = (537, 12, 999)          # one per graph of the small batch
TIME_EMB_DIM = 8
STEPS = 6


def build(ref, **over):
    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    cfg.update(over)
    model = ref.ScorePosNet3D(shims.EasyDict(cfg), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    if cfg.get('time_emb_dim', 0) > 0 and cfg['time_emb_mode'] == 'simple':
        sd = weights.time_emb_state_dict(SEED)
    else:
        sd = weights.make_state_dict(SEED, cfg)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    learnable = {k for k, p in model.named_parameters() if p.requires_grad}
    if cfg.get('time_emb_mode') != 'sin':
        assert learnable <= set(sd), learnable - set(sd)
    return model.eval()


def gen_forward_time(ref, mode):
    model = build(ref, time_emb_dim=TIME_EMB_DIM, time_emb_mode=mode)
    b, lpos, lv = small_batch()
    ppos, lposc, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    t = torch.tensor(TIME_STEPS, dtype=torch.long)
    with torch.no_grad():
        p = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch, time_step=t,
                  return_all=True)
    _save(os.path.join(GOLDEN_DIR, f'forward_time_{mode}.npz'), time_step=t.numpy(), time_emb_dim=np.int64(TIME_EMB_DIM),
          protein_pos=ppos.numpy(), ligand_pos=lposc.numpy(), ligand_v=lv.numpy(),
          pred_ligand_pos=p['pred_ligand_pos'].numpy(), pred_ligand_v=p['pred_ligand_v'].numpy(),
          final_ligand_h=p['final_ligand_h'].numpy(), layer0_pred_ligand_v=p['layer_pred_ligand_v'][0].numpy())
    print(f'forward_time_{mode}: |pred_v| max', float(p['pred_ligand_v'].abs().max()))


def gen_forward_ln_signs(ref):
    model = ref.ScorePosNet3D(shims.EasyDict(dict(weights.DEFAULT_MODEL_CONFIG)), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    res = model.load_state_dict(weights.ln_signs_state_dict(SEED), strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    model.eval()
    b, lpos, lv = small_batch()
    ppos, lposc, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    with torch.no_grad():
        p = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch, return_all=True)
    _save(os.path.join(GOLDEN_DIR, 'forward_ln_signs.npz'), protein_pos=ppos.numpy(), ligand_pos=lposc.numpy(), ligand_v=lv.numpy(),
          pred_ligand_pos=p['pred_ligand_pos'].numpy(), pred_ligand_v=p['pred_ligand_v'].numpy(),
          final_ligand_h=p['final_ligand_h'].numpy(), final_h=p['final_h'].numpy(),
          layer0_pred_ligand_v=p['layer_pred_ligand_v'][0].numpy(), layer0_pred_ligand_pos=p['layer_pred_ligand_pos'][0].numpy())
    print('forward_ln_signs: |pred_v| max', float(p['pred_ligand_v'].abs().max()), '|dx| max',
          float((p['pred_ligand_pos'] - lposc).abs().max()))


def gen_forward_options(ref, name, **over):
    model = build(ref, **over)
    b, lpos, lv = small_batch()
    ppos, lposc, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    with torch.no_grad():
        p = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch)
        f = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch, fix_x=True)
    _save(os.path.join(GOLDEN_DIR, name), protein_pos=ppos.numpy(), ligand_pos=lposc.numpy(), ligand_v=lv.numpy(),
          pred_ligand_pos=p['pred_ligand_pos'].numpy(), pred_ligand_v=p['pred_ligand_v'].numpy(),
          final_ligand_h=p['final_ligand_h'].numpy(), final_h=p['final_h'].numpy(), fix_x_final_ligand_h=f['final_ligand_h'].numpy())
    print(name, over, '|pred_v| max', float(p['pred_ligand_v'].abs().max()), '|dx| max', float((p['pred_ligand_pos'] - lposc).abs().max()))


def gen_forward_blocks(ref):
    model = build(ref, num_blocks=2)
    b, lpos, lv = small_batch()
    ppos, lposc, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    with torch.no_grad():
        p = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch)
        f = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch, fix_x=True)
    _save(os.path.join(GOLDEN_DIR, 'forward_blocks2.npz'), protein_pos=ppos.numpy(), ligand_pos=lposc.numpy(), ligand_v=lv.numpy(),
          pred_ligand_pos=p['pred_ligand_pos'].numpy(), pred_ligand_v=p['pred_ligand_v'].numpy(),
          final_ligand_h=p['final_ligand_h'].numpy(), final_h=p['final_h'].numpy(), fix_x_final_ligand_h=f['final_ligand_h'].numpy())
    print('forward_blocks2: |pred_v| max', float(p['pred_ligand_v'].abs().max()))


def gen_sample(ref, name, base, **over):
    model = build(ref, **over)
    b, lpos, lv = small_batch()
    with counter_draws(base), torch.no_grad():
        r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                                   b.ligand_element_batch, num_steps=STEPS, center_pos_mode='protein')
    _save(os.path.join(GOLDEN_DIR, name), draws_base=np.int64(base), steps=np.int64(STEPS), init_ligand_pos=lpos.numpy(),
          init_ligand_v=lv.numpy(), pos_traj=np.stack([x.numpy() for x in r['pos_traj']]),
          v_traj=np.stack([x.numpy() for x in r['v_traj']]), v0_traj=np.stack([x.numpy() for x in r['v0_traj']]),
          pos=r['pos'].numpy(), v=r['v'].numpy())
    print(name, 'final pos std', float(r['pos'].std()))


def main():
    ref = reference_loader.load()
    gen_forward_time(ref, 'simple')
    # the reference's own 'sin' mode is dead code (its default-initialised module, forward on the small batch)
    cfg = dict(weights.DEFAULT_MODEL_CONFIG, time_emb_dim=TIME_EMB_DIM, time_emb_mode='sin')
    sin_model = ref.ScorePosNet3D(shims.EasyDict(cfg), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM).eval()
    b, lpos, lv = small_batch()
    try:
        with torch.no_grad():
            sin_model(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv, b.ligand_element_batch,
                      time_step=torch.tensor(TIME_STEPS, dtype=torch.long))
        raise SystemExit("time_emb_mode='sin' ran in the reference: the mirror's refusal needs revisiting")
    except RuntimeError as exc:
        assert 'Sizes of tensors must match' in str(exc), exc
        print("reference, time_emb_mode='sin' raises:", str(exc).splitlines()[0])
    gen_sample(ref, 'sample_time_simple_6.npz', 4300, time_emb_dim=TIME_EMB_DIM, time_emb_mode='simple')
    gen_sample(ref, 'sample_noise_6.npz', 4400, model_mean_type='noise')
    gen_forward_blocks(ref)
    gen_forward_ln_signs(ref)
    global STEPS
    STEPS = 4
    gen_sample(ref, 'sample_blocks2_4.npz', 4500, num_blocks=2)
    gen_forward_options(ref, 'forward_ew_r_out_fc.npz', ew_net_type='r', x2h_out_fc=True)
    gen_forward_options(ref, 'forward_ew_none.npz', ew_net_type='none')
    gen_forward_options(ref, 'forward_out_fc.npz', x2h_out_fc=True)
    gen_forward_options(ref, 'forward_ew_m.npz', ew_net_type='m')
    gen_forward_options(ref, 'forward_sync_twoup.npz', sync_twoup=True)
    gen_forward_options(ref, 'forward_stages_2_2.npz', num_x2h=2, num_h2x=2)
    gen_forward_options(ref, 'forward_stages_1_3_r.npz', num_x2h=1, num_h2x=3, ew_net_type='r', x2h_out_fc=True)
    gen_sample(ref, 'sample_ew_r_out_fc_4.npz', 4600, ew_net_type='r', x2h_out_fc=True)


if __name__ == '__main__':
    main()
