"""Round-6 fixtures from the REAL reference (TEST INFRASTRUCTURE; build container only, needs /root/reference):

    PYTHONDONTWRITEBYTECODE=1 python -m oracle.make_golden_r6 [--keep-existing]

Weight regimes the seeded U(-1/sqrt(fan_in), 1/sqrt(fan_in)) sets never reach (no checkpoint ships with the reference, so these stand in
for what a trained / pruned net looks like to the product's folded edge MLPs, csrc/pack.cpp FoldedMlp; models/common.py:60-80):

  forward_ln_dead.npz      oracle.weights.ln_dead_state_dict: in every MLP LayerNorm weights that are exactly 0 with LayerNorm biases of
                           -1 / +1 / +3, tiny weights with a bias of 0.5, negative weights, and the first Linear scaled by 4 -- forward
                           (return_all) on the small batch.  This is the set on which the round-5 fold overflowed (bias / |weight|).
  forward_trained_g{4,8}.npz   oracle.weights.trained_like_state_dict(gain): every Linear of the MLPs at 4 / 8 times nn.Linear's range,
                           LayerNorm weights log-uniform in [0.05, 5], biases uniform in [-2, 2] -- forward (return_all) on the small
                           batch; per-layer h / x through hooks on the real layers.
  sample_trained_g{4,8}_20.npz   20 reverse steps of the reference's own loop (models/molopt_score_model.py:633-703) with those weights
                           and the counter draws: every step's positions and types, for teacher-forced single steps on the GPU.
  Each forward fixture also holds the same reference run in float64 (``*_f64``: the module cast to double, ``Tensor.float`` patched to
  ``double`` for the run so that the one-hot inputs follow).  The fp32 reference itself is 3e-6 (gain 4) / 2e-4 (gain 8) away from it, which
  is what the tolerance of the GPU tests on these sets is stated against (tests/test_gpu_weight_regimes.py).
The fixtures hold inputs and outputs only; weights are regenerated from the seed.
"""
from __future__ import annotations

import contextlib
import os

import numpy as np
import torch

from . import reference_loader, shims, weights
from .make_golden import GOLDEN_DIR, SEED, _save, small_batch
from .make_golden_r2 import counter_draws

STEPS = 20


def build(ref, sd):
    model = ref.ScorePosNet3D(shims.EasyDict(dict(weights.DEFAULT_MODEL_CONFIG)), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    res = model.load_state_dict(sd, strict=False)
    assert not res.unexpected_keys, res.unexpected_keys
    return model.eval()


@contextlib.contextmanager
def float64_run():
    """The reference casts its one-hot inputs with ``.float()`` (models/molopt_score_model.py:319-338): evaluate everything in double."""
    orig, dflt = torch.Tensor.float, torch.get_default_dtype()
    torch.Tensor.float = lambda self, *a, **k: self.double()
    torch.set_default_dtype(torch.float64)
    try:
        yield
    finally:
        torch.Tensor.float = orig
        torch.set_default_dtype(dflt)


def gen_forward(ref, name, sd):
    model = build(ref, sd)
    b, lpos, lv = small_batch()
    ppos, lposc, _ = ref.center_pos(b.protein_pos, lpos, b.protein_element_batch, b.ligand_element_batch, mode='protein')
    per_layer = {'h': [], 'x': []}

    def hook(m, i, o):
        per_layer['h'].append(o[0].detach().clone())
        per_layer['x'].append(o[1].detach().clone())
    hooks = [layer.register_forward_hook(hook) for layer in model.refine_net.base_block]
    with torch.no_grad():
        p = model(ppos, b.protein_atom_feature.float(), b.protein_element_batch, lposc, lv, b.ligand_element_batch, return_all=True)
    for h in hooks:
        h.remove()
    with float64_run(), torch.no_grad():
        p64 = build(ref, sd).double()(ppos.double(), b.protein_atom_feature.double(), b.protein_element_batch, lposc.double(), lv,
                                      b.ligand_element_batch)
    assert p64['final_h'].dtype == torch.float64
    _save(os.path.join(GOLDEN_DIR, name), protein_pos=ppos.numpy(), ligand_pos=lposc.numpy(), ligand_v=lv.numpy(),
          pred_ligand_pos=p['pred_ligand_pos'].numpy(), pred_ligand_v=p['pred_ligand_v'].numpy(),
          final_ligand_h=p['final_ligand_h'].numpy(), final_h=p['final_h'].numpy(),
          layer0_pred_ligand_v=p['layer_pred_ligand_v'][0].numpy(), layer0_pred_ligand_pos=p['layer_pred_ligand_pos'][0].numpy(),
          h_layers=torch.stack(per_layer['h']).numpy(), x_layers=torch.stack(per_layer['x']).numpy(),
          pred_ligand_pos_f64=p64['pred_ligand_pos'].numpy(), pred_ligand_v_f64=p64['pred_ligand_v'].numpy(), final_h_f64=p64['final_h'].numpy())
    print('   fp32 reference vs its float64 run:', {k: float((p[k].double() - p64[k]).abs().max()) for k in ('pred_ligand_pos', 'pred_ligand_v', 'final_h')})
    print(f'{name}: |pred_v| max {float(p["pred_ligand_v"].abs().max()):.3f}  |dx| max {float((p["pred_ligand_pos"] - lposc).abs().max()):.3f}'
          f'  |h| max {float(p["final_h"].abs().max()):.2f}')


def gen_sample(ref, name, base, sd):
    model = build(ref, sd)
    b, lpos, lv = small_batch()
    with counter_draws(base), torch.no_grad():
        r = model.sample_diffusion(b.protein_pos, b.protein_atom_feature.float(), b.protein_element_batch, lpos, lv,
                                   b.ligand_element_batch, num_steps=STEPS, center_pos_mode='protein')
    _save(os.path.join(GOLDEN_DIR, name), draws_base=np.int64(base), steps=np.int64(STEPS), init_ligand_pos=lpos.numpy(),
          init_ligand_v=lv.numpy(), pos_traj=np.stack([x.numpy() for x in r['pos_traj']]),
          v_traj=np.stack([x.numpy() for x in r['v_traj']]), v0_traj=np.stack([x.numpy() for x in r['v0_traj']]),
          pos=r['pos'].numpy(), v=r['v'].numpy())
    print(name, 'final pos std', float(r['pos'].std()), 'types changed per step',
          [int((r['v_traj'][i] != r['v_traj'][i - 1]).sum()) for i in range(1, STEPS)])


def main():
    ref = reference_loader.load()
    torch.set_num_threads(8)
    gen_forward(ref, 'forward_ln_dead.npz', weights.ln_dead_state_dict(SEED))
    for gain, base in ((4, 6100), (8, 6200)):
        sd = weights.trained_like_state_dict(SEED, float(gain))
        gen_forward(ref, f'forward_trained_g{gain}.npz', sd)
        gen_sample(ref, f'sample_trained_g{gain}_20.npz', base, sd)


if __name__ == '__main__':
    main()
