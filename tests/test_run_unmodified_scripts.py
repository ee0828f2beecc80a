"""`python -m targetdiff_amd.run scripts/sample_diffusion.py ...`: the reference's sampling script, as a file and unmodified, executed
as __main__ with `models.molopt_score_model.ScorePosNet3D` bound to the mirror class (build container only: needs /root/reference).
The native layer is the recording stub of test_reference_driver_with_mirror (no GPU here); what is checked is the seam -- the
script constructs, loads, moves and drives the mirror exactly as it does its own class, and writes result_{id}.pt."""
import os
import sys
import types

import numpy as np
import pytest
import torch

from oracle import reference_loader

pytestmark = pytest.mark.skipif(not reference_loader.available(), reason='reference tree not present (GPU box)')


def test_reference_sampling_script_runs_unmodified_on_the_mirror(tmp_path, monkeypatch):
    from oracle import shims, weights
    from oracle.shims import EasyDict
    from oracle.make_golden import SEED
    from oracle.make_golden_r2 import counter_draws, driver_data
    from test_reference_driver_with_mirror import RecordingNative, StubSession
    from targetdiff_amd import capi, models, run
    shims.install()                                            # torch_geometric / torch_scatter stand-ins (not installed here)
    # what only the script's __main__ uses and this container lacks (rdkit, lmdb): transforms and the dataset
    tr = types.ModuleType('utils.transforms')

    class _Feat:
        def __init__(self, *a, dim=None):
            self._dim = dim

        @property
        def feature_dim(self):
            return self._dim

        def __call__(self, data):
            return data
    tr.FeaturizeProteinAtom = lambda: _Feat(dim=weights.PROTEIN_FEATURE_DIM)
    tr.FeaturizeLigandAtom = lambda mode: _Feat(dim=weights.LIGAND_FEATURE_DIM)
    tr.FeaturizeLigandBond = lambda: _Feat()
    import importlib
    if reference_loader.REFERENCE_ROOT not in sys.path:
        monkeypatch.syspath_prepend(reference_loader.REFERENCE_ROOT)
    importlib.import_module('utils')
    monkeypatch.setitem(sys.modules, 'utils.transforms', tr)
    ds = types.ModuleType('datasets')
    ds.get_dataset = lambda config, transform: (None, {'train': [], 'test': [driver_data(), driver_data()]})
    pl = types.ModuleType('datasets.pl_data')
    pl.FOLLOW_BATCH = ('protein_element', 'ligand_element', 'ligand_bond_type',)           # datasets/pl_data.py:7
    ds.pl_data = pl
    monkeypatch.setitem(sys.modules, 'datasets', ds)
    monkeypatch.setitem(sys.modules, 'datasets.pl_data', pl)

    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    # a checkpoint as the reference's training loop writes it: the full state_dict of the reference's own class (384 keys)
    refmod = reference_loader.load()
    ref_cls = getattr(refmod, '_reference_ScorePosNet3D', None) or refmod.ScorePosNet3D
    ref_model = ref_cls(EasyDict(cfg), weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    ref_model.load_state_dict(weights.make_state_dict(SEED), strict=False)
    sd = {k: v.clone() for k, v in ref_model.state_dict().items()}
    assert len(sd) == 384
    ckpt = {'config': EasyDict(model=cfg, data=EasyDict(transform=EasyDict(ligand_atom_mode='add_aromatic'))), 'model': sd}
    ckpt_path = tmp_path / 'model.pt'
    torch.save({'model': sd}, ckpt_path)                       # (the script's torch.load is routed to `ckpt` below: EasyDict pickles)
    monkeypatch.setattr(torch, 'load', lambda *a, **k: ckpt)
    yml = tmp_path / 'sampling.yml'
    yml.write_text(f'model:\n  checkpoint: {ckpt_path}\nsample:\n  seed: 2021\n  num_samples: 3\n  num_steps: 2\n  pos_only: False\n'
                   '  center_pos_mode: protein\n  sample_num_atoms: prior\n')
    log = []
    native = RecordingNative(sd, cfg, weights.LIGAND_FEATURE_DIM, log)
    monkeypatch.setattr(models.ScorePosNet3D, '_native', lambda self, device: native)
    monkeypatch.setattr(capi, 'NativeSession', StubSession)
    made = []
    orig_init = models.ScorePosNet3D.__init__

    def spy_init(self, *a, **k):
        made.append(self)
        orig_init(self, *a, **k)
    monkeypatch.setattr(models.ScorePosNet3D, '__init__', spy_init)
    out = tmp_path / 'out'
    script = os.path.join(reference_loader.REFERENCE_ROOT, 'scripts', 'sample_diffusion.py')
    before = open(script, 'rb').read()
    monkeypatch.setattr(sys, 'argv', list(sys.argv))
    sys.modules.pop('scripts.sample_diffusion', None)
    with counter_draws(3300, lambda kind, n: n):
        run.main(['--reference-root', reference_loader.REFERENCE_ROOT, 'scripts/sample_diffusion.py', str(yml), '-i', '1', '--device', 'cpu',
                  '--batch_size', '2', '--result_path', str(out)])
    assert open(script, 'rb').read() == before
    assert sys.modules['models.molopt_score_model'].ScorePosNet3D is models.ScorePosNet3D
    run.uninstall()                    # other tests of this process use the reference's own class
    # the script built ONE model, and it is the mirror; it loaded the checkpoint strictly and drove it through both batches
    assert len(made) == 1 and type(made[0]) is models.ScorePosNet3D
    names = [n for n, _ in log]
    assert names.count('session_create') == 2 and names.count('model_forward') == 2 * 2 == names.count('posterior_step')
    monkeypatch.undo()
    res = torch.load(out / 'result_1.pt', weights_only=False)
    assert set(res) == {'data', 'pred_ligand_pos', 'pred_ligand_v', 'pred_ligand_pos_traj', 'pred_ligand_v_traj', 'time'}
    assert len(res['pred_ligand_pos']) == 3 and res['pred_ligand_pos'][0].dtype == np.float64
    assert res['pred_ligand_pos_traj'][0].shape[0] == 2 and len(res['time']) == 2
    assert (out / 'sample.yml').exists()


def test_install_is_idempotent_and_keeps_the_reference_class_reachable(monkeypatch):
    from oracle import shims
    from targetdiff_amd import models, run
    shims.install()
    ref = run.install(reference_loader.REFERENCE_ROOT)
    assert ref.ScorePosNet3D is models.ScorePosNet3D
    kept = ref._reference_ScorePosNet3D
    assert kept is not None and kept is not models.ScorePosNet3D and kept.__module__ == 'models.molopt_score_model'
    assert run.install() is ref and ref._reference_ScorePosNet3D is kept
    run.uninstall()                   # leave the reference module as other tests expect it
    assert ref.ScorePosNet3D is kept and not hasattr(ref, '_reference_ScorePosNet3D')
