"""tests/golden/ligand_1h36_docked.npz: the docked ligand of the reference's example pocket (build container only).

    python -m oracle.make_docked_fixture

Reads /root/reference/examples/1h36_A_rec_1h36_r88_lig_tt_docked_0.sdf (the ligand the reference's README samples next to,
`scripts/sample_for_pocket.py --pdb_path examples/1h36_A_rec_1h36_r88_lig_tt_docked_0_pocket10.pdb`) -- coordinates in the
frame of tests/golden/pocket_1h36.npz -- and writes positions + element symbols.  bench.py's geometry sweep replicates this pose
(the geometry a trained model's trajectory ends in: a compact ligand docked inside the pocket) next to the Gaussian clouds.
TEST / BENCH INFRASTRUCTURE; the product never reads it."""
import os

import numpy as np

from . import reference_loader

SDF = 'examples/1h36_A_rec_1h36_r88_lig_tt_docked_0.sdf'
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden', 'ligand_1h36_docked.npz')


def parse_sdf(path):
    lines = open(path).read().splitlines()
    counts = lines[3]
    n_atoms = int(counts[0:3])
    pos, sym = [], []
    for line in lines[4:4 + n_atoms]:
        pos.append([float(line[0:10]), float(line[10:20]), float(line[20:30])])
        sym.append(line[31:34].strip())
    return np.asarray(pos, np.float32), sym


def main():
    pos, sym = parse_sdf(os.path.join(reference_loader.REFERENCE_ROOT, SDF))
    np.savez(OUT, pos=pos, elements=np.asarray(sym), source=np.asarray(SDF))
    print(OUT, pos.shape, sorted(set(sym)), 'centroid', pos.mean(0))


if __name__ == '__main__':
    main()
