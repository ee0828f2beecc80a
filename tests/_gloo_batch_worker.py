"""Rank body of tests/test_launcher_gloo.py: tools/batch_sample.py's main() on CPU (gloo) with a stand-in model.  Started
by targetdiff_amd.launch.spawn_ranks, i.e. it sees the same RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* environment a GPU rank
does."""
import importlib.util
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


class StubModel:
    """The two members the driver touches (num_classes, sample_diffusion); deterministic output per pocket."""
    num_classes = 13

    def sample_diffusion(self, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, batch_ligand,
                         num_steps=None, center_pos_mode=None, pos_only=False, max_graph_nodes=0):
        steps = num_steps or 2
        n = init_ligand_pos.shape[0]
        base = protein_pos.mean(0, keepdim=True).expand(n, 3).clone()
        return {'pos': base, 'v': torch.zeros(n, dtype=torch.long),
                'pos_traj': [base.clone() for _ in range(steps)], 'v_traj': [torch.zeros(n, dtype=torch.long)] * steps,
                'v0_traj': [torch.zeros(n, 13)] * steps, 'vt_traj': [torch.zeros(n, 13)] * steps}


def main():
    spec = importlib.util.spec_from_file_location('batch_sample', os.path.join(ROOT, 'tools', 'batch_sample.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(sys.argv[1:], model_factory=lambda args, dev: StubModel())


if __name__ == '__main__':
    main()
