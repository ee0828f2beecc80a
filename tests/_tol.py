"""Stated tolerances of the GPU parity tests, and a comparison helper that records how much of each tolerance a test used.

SURVEY.md section 7 asks for |dx| <= 1e-5 A and |dh| <= 1e-4 on teacher-forced steps; a single forward pass against a golden of the real
reference is held to 5e-6 (measured: <= 1.9e-6 A / 8.6e-7), free-running trajectories to 5e-5 A (they amplify the k-NN graph's
discontinuities; measured 2.7e-5 at the end of 1000 steps).  Neighbour indices, sampled atom types and session-vs-stateless
comparisons are exact (torch.equal) and do not pass through here.

Every call of `close` is recorded; tests/conftest.py prints the table of margins (difference / tolerance) at the end of the run and,
on the GPU box, writes it to gpurun_out/margins.txt.
"""
import inspect
import os

import numpy as np
import torch

TOL_X = 1e-5          # A, teacher-forced steps and multi-stage comparisons
TOL_H = 1e-4          # features / logits / log-probabilities, same
TOL_FWD = 5e-6        # one forward pass against a reference golden (positions in A and features alike)
TOL_TRAJ = 5e-5       # A, free-running trajectories

MARGINS = []          # (test, where, difference, tolerance)


def maxdiff(a, b):
    a = a.detach().cpu().double().numpy() if torch.is_tensor(a) else np.asarray(a, np.float64)
    b = b.detach().cpu().double().numpy() if torch.is_tensor(b) else np.asarray(b, np.float64)
    return float(np.max(np.abs(a - b))) if a.size else 0.0


def close(a, b, tol, what=None):
    """assert max |a - b| <= tol, recording the margin"""
    d = maxdiff(a, b)
    test = os.environ.get('PYTEST_CURRENT_TEST', '?').split(' ')[0].split('::')[-1]
    frame = inspect.stack()[1]
    where = f'{os.path.basename(frame.filename)}:{frame.lineno}' + (f' {what}' if what is not None else '')
    MARGINS.append((test, where, d, float(tol)))
    assert d <= tol, f'{where}: max |difference| {d:.3e} > tolerance {tol:.1e}'
    return d
