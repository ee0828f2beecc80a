import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')
    # The oracle's small torch-CPU ops do not scale past a few threads, and on an 8-vCPU container 8 OpenMP threads next to
    # the launcher tests' child processes degrade into spinning (the CPU suite then takes 6 min instead of 1.5): cap at 4.
    torch.set_num_threads(max(1, min(4, os.cpu_count() or 1)))


def load_golden(name):
    with np.load(os.path.join(GOLDEN, name)) as z:
        return {k: z[k] for k in z.files}


@pytest.fixture(scope='session')
def state_dict():
    from oracle import weights
    return weights.make_state_dict(2021)


@pytest.fixture(scope='session')
def golden_small():
    return load_golden('forward_small.npz')


def small_inputs(g):
    """torch tensors of the forward_small fixture inputs (already centred)."""
    return dict(
        protein_pos=torch.from_numpy(g['protein_pos']),
        protein_v=torch.from_numpy(g['protein_feat'].astype(np.float32)),
        batch_protein=torch.from_numpy(g['batch_protein']),
        ligand_pos=torch.from_numpy(g['ligand_pos']),
        ligand_v=torch.from_numpy(g['ligand_v']),
        batch_ligand=torch.from_numpy(g['batch_ligand']),
    )


def pocket_1h36():
    from targetdiff_amd import workloads
    g = load_golden('pocket_1h36.npz')
    return workloads.Pocket(g['pos'], g['feat'].astype(np.int64), '1h36_pocket10'), g['prior_sizes_seed2021']


def pytest_terminal_summary(terminalreporter):
    """How much of its stated tolerance every comparison used (tests/_tol.py): the largest first."""
    try:
        from _tol import MARGINS
    except Exception:
        return
    if not MARGINS:
        return
    rows = sorted(MARGINS, key=lambda r: -(r[2] / r[3] if r[3] > 0 else 0.0))
    lines = [f'{d / tol if tol > 0 else 0.0:6.3f}  {d:10.3e} / {tol:8.1e}  {test}  {where}' for test, where, d, tol in rows]
    worst = {}          # the tightest comparison of every test
    for test, where, d, tol in rows:
        worst.setdefault(test, (d, tol, where))
    terminalreporter.write_line(f'tolerance margins (difference / tolerance): {len(rows)} comparisons in {len(worst)} tests, the tightest of each test, largest first:')
    for test, (d, tol, where) in list(worst.items())[:20]:
        terminalreporter.write_line(f'  {d / tol if tol > 0 else 0.0:6.3f}  {d:10.3e} / {tol:8.1e}  {test}  {where}')
    out = os.path.join(ROOT, 'gpurun_out')
    if os.path.isdir(out):
        with open(os.path.join(out, 'margins.txt'), 'w') as f:
            f.write('ratio   difference / tolerance   test   where\n' + '\n'.join(lines) + '\n')
