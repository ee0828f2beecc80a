"""World-size-2 gloo run of the pocket-sharded driver (the N > 1 path of bench.py / sampling.run_sharded) on CPU with a
stand-in model: checks the partition, that no data-path collective is needed, and the metadata gather."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


class _StubModel:
    """Has the two members sample_diffusion_ligand touches: num_classes and sample_diffusion."""
    num_classes = 13

    def sample_diffusion(self, protein_pos, protein_v, batch_protein, init_ligand_pos, init_ligand_v, batch_ligand,
                         num_steps=None, center_pos_mode=None, pos_only=False, max_graph_nodes=0):
        n = init_ligand_pos.shape[0]
        steps = num_steps or 2
        return {'pos': init_ligand_pos + 1.0, 'v': init_ligand_v,
                'pos_traj': [init_ligand_pos.clone() for _ in range(steps)],
                'v_traj': [init_ligand_v.clone() for _ in range(steps)],
                'v0_traj': [torch.zeros(n, 13) for _ in range(steps)],
                'vt_traj': [torch.zeros(n, 13) for _ in range(steps)]}


def _worker(rank, world, port, out):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from targetdiff_amd import sampling, workloads
    pockets = [workloads.synthetic_pocket(100 + i, 12 + i) for i in range(5)]
    res = sampling.run_sharded(_StubModel(), pockets, num_samples=3, rank=rank, world_size=world, batch_size=2,
                               device='cpu', num_steps=2, ligand_num_atoms=[4, 5, 6])
    meta = {'rank': rank, 'pockets': sorted(res), 'ligands': sum(len(r[0]) for r in res.values())}
    gathered = sampling.gather_metadata(meta)
    dist.barrier()
    if rank == 0:
        torch.save({'gathered': gathered, 'shapes': {k: [p.shape for p in v[0]] for k, v in res.items()}}, out)
    dist.destroy_process_group()


def test_pocket_sharding_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'r0.pt')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    r = torch.load(out, weights_only=False)
    g = sorted(r['gathered'], key=lambda m: m['rank'])
    assert g[0]['pockets'] == [0, 2, 4] and g[1]['pockets'] == [1, 3]      # i % world == rank
    assert g[0]['ligands'] == 9 and g[1]['ligands'] == 6                   # 3 samples per pocket
    assert r['shapes'][0] == [(4, 3), (5, 3), (6, 3)]


# ---- the census bench.py puts into its JSON line (launch.rank_census): what the process group looks like, gathered from every rank
def _census_worker(rank, world, port, out):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    from targetdiff_amd import launch
    c = launch.rank_census('cpu', seconds_per_step=0.004 * (rank + 1), extra={'pockets': 3 - rank})
    if rank == 0:
        torch.save(c, out)
    dist.barrier()
    dist.destroy_process_group()


def test_rank_census_two_ranks_gloo(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'census.pt')
    mp.spawn(_census_worker, args=(2, port, out), nprocs=2, join=True)
    c = torch.load(out, weights_only=False)
    assert c['world_size'] == 2 and c['env_world_size'] == 2 and c['backend'] == 'gloo'
    assert [r['rank'] for r in c['ranks']] == [0, 1] and [r['local_rank'] for r in c['ranks']] == [0, 1]
    assert [r['pockets'] for r in c['ranks']] == [3, 2] and c['ranks'][0]['pid'] != c['ranks'][1]['pid']
    assert np.allclose(c['ms_per_step_per_rank'], [4.0, 8.0]) and abs(c['max_over_mean'] - 8.0 / 6.0) < 1e-12


def test_rank_census_refuses_two_ranks_on_one_device(monkeypatch):
    """Two ranks that report the same (host, PCI address) are the failure the census exists for: LOCAL_RANK ignored, or a device list
    that maps every rank to GPU 0."""
    import pytest
    from targetdiff_amd import launch

    def fake_gather(dst, obj):
        dst[0] = dict(obj, rank=0)
        dst[1] = dict(obj, rank=1)          # same host, same PCI address
    monkeypatch.setattr(launch, '_device_identity', lambda d: {'device': 'cuda:0', 'host': 'box', 'pid': 1, 'ordinal': 0,
                                                               'pci': '0000:05:00', 'uuid': None})
    monkeypatch.setattr(dist, 'is_initialized', lambda: True)
    monkeypatch.setattr(dist, 'get_world_size', lambda: 2)
    monkeypatch.setattr(dist, 'get_backend', lambda: 'nccl')
    monkeypatch.setattr(dist, 'get_rank', lambda: 0)
    monkeypatch.setattr(dist, 'all_gather_object', fake_gather)
    with pytest.raises(RuntimeError, match='same device'):
        launch.rank_census('cuda:0', 0.004)
    # a census of one, without a process group
    monkeypatch.setattr(dist, 'is_initialized', lambda: False)
    c = launch.rank_census('cuda:0', 0.004)
    assert c['world_size'] == 1 and c['backend'] is None and len(c['ranks']) == 1
