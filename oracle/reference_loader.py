"""Import the REAL reference model code (TEST INFRASTRUCTURE; build container only).

/root/reference does not exist on the GPU box, so nothing that runs there may import this module.
It is used by ``oracle/make_golden.py`` (fixture generation) and by the optional
``tests/test_oracle_vs_reference.py`` (skipped when the reference tree is absent).

The reference files are imported unmodified: models/molopt_score_model.py, models/uni_transformer.py,
models/common.py, models/egnn.py.  Their third-party imports are satisfied by ``oracle.shims``.
"""
from __future__ import annotations

import os
import sys

REFERENCE_ROOT = os.environ.get('TARGETDIFF_REFERENCE', '/root/reference')


def available() -> bool:
    return os.path.isfile(os.path.join(REFERENCE_ROOT, 'models', 'molopt_score_model.py'))


def load():
    """Returns the reference ``models.molopt_score_model`` module."""
    if not available():
        raise RuntimeError(f'reference tree not found at {REFERENCE_ROOT}')
    from . import shims
    shims.install()
    sys.dont_write_bytecode = True          # the reference tree is read-only by contract
    if REFERENCE_ROOT not in sys.path:
        sys.path.insert(0, REFERENCE_ROOT)
    import importlib
    return importlib.import_module('models.molopt_score_model')


def load_driver():
    """Returns the reference ``scripts.sample_diffusion`` module (its ``sample_diffusion_ligand``, :31-116, is the
    batching driver).  Its module-level imports that only ``__main__`` uses (``utils.transforms`` -> rdkit, ``datasets``
    -> lmdb / rdkit) are stubbed; ``utils.misc``, ``utils.evaluation.atom_num`` and the model files are the real ones."""
    import importlib
    import types
    load()
    if 'utils.transforms' not in sys.modules:
        importlib.import_module('utils')
        sys.modules['utils.transforms'] = types.ModuleType('utils.transforms')
    if 'datasets' not in sys.modules:
        ds = types.ModuleType('datasets')
        ds.get_dataset = None
        pl = types.ModuleType('datasets.pl_data')
        pl.FOLLOW_BATCH = ('protein_element', 'ligand_element', 'ligand_bond_type',)      # datasets/pl_data.py:7
        ds.pl_data = pl
        sys.modules['datasets'] = ds
        sys.modules['datasets.pl_data'] = pl
    return importlib.import_module('scripts.sample_diffusion')
