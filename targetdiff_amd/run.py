"""Run an UNMODIFIED reference script on the MI355X path.

    python -m targetdiff_amd.run [--reference-root DIR] scripts/sample_diffusion.py configs/sampling.yml -i 0 --batch_size 100
    python -m targetdiff_amd.run scripts/sample_for_pocket.py configs/sampling.yml --pdb_path examples/...pdb

The reference scripts bind the model with ``from models.molopt_score_model import ScorePosNet3D``
(scripts/sample_diffusion.py:17, scripts/sample_for_pocket.py:11, scripts/likelihood_est_diffusion.py).  This runner
imports the reference's own ``models.molopt_score_model`` first -- its helpers (``log_sample_categorical`` ...), datasets,
transforms and utilities stay the reference's -- replaces the one attribute ``ScorePosNet3D`` on it with the HIP-backed
mirror (targetdiff_amd.models.ScorePosNet3D: same constructor, ``state_dict`` keys, ``forward`` / ``sample_diffusion`` /
``likelihood_estimation`` / ``fetch_embedding`` signatures and return dictionaries) and then executes the script file,
untouched, as ``__main__`` with the remaining command line.  No file of the reference tree is edited or copied.

``install()`` does the attribute swap alone, for callers that import the reference's modules themselves.
"""
from __future__ import annotations

import importlib
import os
import runpy
import sys


def install(reference_root: str | None = None):
    """Make ``models.molopt_score_model.ScorePosNet3D`` (of the reference tree on sys.path, or under ``reference_root``) the
    HIP-backed mirror; returns the patched reference module.  Idempotent."""
    if reference_root:
        reference_root = os.path.abspath(reference_root)
        if reference_root not in sys.path:
            sys.path.insert(0, reference_root)
    try:
        ref = importlib.import_module('models.molopt_score_model')
    except ImportError as exc:
        raise ImportError('targetdiff_amd.run: cannot import the reference\'s models.molopt_score_model -- run from the root of '
                          'the reference checkout or pass --reference-root (its own dependencies, torch_geometric / '
                          f'torch_scatter, must be importable): {exc}') from exc
    from .models import ScorePosNet3D, get_refine_net
    if getattr(ref, 'ScorePosNet3D', None) is not ScorePosNet3D:
        ref._reference_ScorePosNet3D = getattr(ref, 'ScorePosNet3D', None)       # still reachable, e.g. for A/B runs
        ref._reference_get_refine_net = getattr(ref, 'get_refine_net', None)
        ref.ScorePosNet3D = ScorePosNet3D
        ref.get_refine_net = get_refine_net
    return ref


def uninstall() -> None:
    """Undo :func:`install` (the reference module gets its own class back)."""
    ref = sys.modules.get('models.molopt_score_model')
    if ref is not None and getattr(ref, '_reference_ScorePosNet3D', None) is not None:
        ref.ScorePosNet3D = ref._reference_ScorePosNet3D
        ref.get_refine_net = ref._reference_get_refine_net
        del ref._reference_ScorePosNet3D, ref._reference_get_refine_net


def main(argv=None) -> None:
    argv = list(sys.argv[1:] if argv is None else argv)
    root = None
    if argv and argv[0] == '--reference-root':
        if len(argv) < 2:
            raise SystemExit('usage: python -m targetdiff_amd.run [--reference-root DIR] <script.py> [script arguments ...]')
        root, argv = argv[1], argv[2:]
    if not argv or argv[0] in ('-h', '--help'):
        raise SystemExit('usage: python -m targetdiff_amd.run [--reference-root DIR] <script.py> [script arguments ...]')
    script = argv[0]
    if not os.path.isfile(script) and root and os.path.isfile(os.path.join(root, script)):
        script = os.path.join(root, script)
    if not os.path.isfile(script):
        raise SystemExit(f'targetdiff_amd.run: {argv[0]}: no such script')
    if root is None:
        # scripts/<name>.py inside a checkout: the checkout is the directory above (the reference is run from there)
        here = os.path.dirname(os.path.abspath(script))
        cand = os.path.dirname(here)
        root = cand if os.path.isfile(os.path.join(cand, 'models', 'molopt_score_model.py')) else os.getcwd()
    install(root)
    sys.argv = [script] + argv[1:]
    runpy.run_path(script, run_name='__main__')


if __name__ == '__main__':
    main()
