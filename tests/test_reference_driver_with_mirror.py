"""The import swap of INTEGRATION.md, executed: the REFERENCE's own batching driver (scripts/sample_diffusion.py:31-116,
imported unmodified) drives ``targetdiff_amd.models.ScorePosNet3D`` -- the class a maintainer swaps in at
scripts/sample_diffusion.py:17 -- through its whole call sequence: Batch.from_data_list, model.num_classes, the keyword call of
``model.sample_diffusion``, the un-batching of the returned CPU trajectories.

There is no GPU in the build container and no /root/reference on the GPU box, so the native layer is replaced here by a
RECORDING STUB with capi's method surface whose arithmetic comes from the oracle restatement (test infrastructure).  What this
pins is the seam: argument names / order / dtypes / devices the reference passes, the members it reads, the dictionary keys,
list lengths, dtypes and de-centring it expects back.  Because the restatement is pinned to the reference, the driver's
7-tuple must equal the fixture the reference's driver produced with the reference's own model (tests/golden/driver_small.npz).
CPU only; skipped where the reference tree is absent."""
import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import reference_loader, weights
from oracle import restatement as R

pytestmark = pytest.mark.skipif(not reference_loader.available(), reason='reference tree not present')


def _batch_of(ptr):
    n = (ptr[1:] - ptr[:-1]).long()
    return torch.repeat_interleave(torch.arange(n.numel()), n)


class RecordingNative:
    """capi.NativeModel's surface on torch CPU tensors; every call is logged as (name, shapes / dtypes of the arguments)."""

    def __init__(self, sd, cfg, num_classes, log):
        self.sd, self.cfg, self.num_classes, self.log = sd, cfg, num_classes, log
        self.sched = R.diffusion_schedules(cfg)

    def _rec(self, name, **tensors):
        self.log.append((name, {k: (tuple(v.shape), str(v.dtype), v.device.type) for k, v in tensors.items() if torch.is_tensor(v)}))

    def graph_ptr(self, batch, B):
        self._rec('graph_ptr', batch=batch)
        assert batch.dtype == torch.int64
        return torch.cat([torch.zeros(1, dtype=torch.int64), torch.bincount(batch, minlength=B).cumsum(0)]).to(torch.int32)

    def center_pos(self, protein_pos, protein_ptr, ligand_pos, ligand_ptr, offset=None, sign=-1):
        self._rec('center_pos', protein_pos=protein_pos, ligand_pos=ligand_pos)
        assert offset is None and sign == -1
        bp, bl = _batch_of(protein_ptr), _batch_of(ligand_ptr)
        p, l, off = R.center_positions(protein_pos, ligand_pos, bp, bl)
        protein_pos.copy_(p)
        ligand_pos.copy_(l)
        return off

    def model_forward(self, protein_pos, protein_v, protein_ptr, ligand_pos, ligand_v, ligand_ptr, fix_x=False, max_graph_nodes=0,
                      want_final_h=True, out=None, ligand_graph_bias=None):
        self._rec('model_forward', protein_pos=protein_pos, protein_v=protein_v, ligand_pos=ligand_pos, ligand_v=ligand_v)
        assert protein_v.dtype == torch.float32 and ligand_v.dtype == torch.int64
        return R.model_forward(self.sd, self.cfg, protein_pos, protein_v, _batch_of(protein_ptr), ligand_pos, ligand_v,
                               _batch_of(ligand_ptr), fix_x=fix_x)

    def posterior_step(self, t, ligand_ptr, ligand_pos, ligand_v, pred_pos, pred_v, noise, uniform, pos_next=None, v_next=None,
                       log_v0=None, log_post=None):
        self._rec('posterior_step', t=t, noise=noise, uniform=uniform)
        assert t.dtype == torch.int32 and t.numel() == ligand_ptr.numel() - 1
        pos, v, l0, lp = R.posterior_step(self.sched, t.long(), ligand_pos, ligand_v, pred_pos, pred_v, _batch_of(ligand_ptr), noise,
                                          uniform, self.num_classes)
        pos_next.copy_(pos)
        v_next.copy_(v)
        if log_v0 is not None:
            log_v0.copy_(l0)
            log_post.copy_(lp)
        return pos_next, v_next


class StubSession:
    """capi.NativeSession: the sampling session is an optimisation of model_forward, bit-identical by construction."""

    def __init__(self, native, protein_pos, protein_v, protein_ptr, ligand_ptr, num_ligand_atoms, max_graph_nodes):
        native._rec('session_create', protein_pos=protein_pos, protein_v=protein_v)
        self.a = (native, protein_pos, protein_v, protein_ptr, ligand_ptr)

    def forward(self, ligand_pos, ligand_v, out=None, ligand_graph_bias=None):
        n, pp, pv, pptr, lptr = self.a
        return n.model_forward(pp, pv, pptr, ligand_pos, ligand_v, lptr)

    # td_session_step: forward + posterior update on the state held in `io` (current positions / types updated in place, slot s of
    # the trajectories filled, the step index advanced)
    def make_step_io(self, step_index, t_all, ligand_pos, ligand_v, noise, uniform, pos_traj, v_traj, v0_traj=None, vt_traj=None,
                     pos_only=False, ligand_graph_bias=None):
        assert ligand_graph_bias is None               # time_emb_dim == 0 in this configuration
        return dict(step=step_index, t_all=t_all, pos=ligand_pos, v=ligand_v, noise=noise, uniform=uniform, pos_traj=pos_traj,
                    v_traj=v_traj, v0_traj=v0_traj, vt_traj=vt_traj, pos_only=pos_only)

    def step(self, io, use_graph=True):
        n, pp, pv, pptr, lptr = self.a
        s = int(io['step'][0])
        preds = n.model_forward(pp, pv, pptr, io['pos'], io['v'], lptr)
        full = not io['pos_only']
        v_next = io['v_traj'][s] if full else torch.empty_like(io['v'])
        n.posterior_step(io['t_all'][s], lptr, io['pos'], io['v'], preds['pred_ligand_pos'], preds['pred_ligand_v'], io['noise'],
                         io['uniform'], pos_next=io['pos_traj'][s], v_next=v_next, log_v0=io['v0_traj'][s] if full else None,
                         log_post=io['vt_traj'][s] if full else None)
        io['pos'].copy_(io['pos_traj'][s])
        if full:
            io['v'].copy_(io['v_traj'][s])
        else:
            io['v_traj'][s].copy_(io['v'])
        io['step'][0] += 1


def test_reference_driver_runs_on_the_mirror_and_reproduces_its_own_outputs(monkeypatch):
    from oracle.make_golden import SEED
    from oracle.make_golden_r2 import counter_draws, driver_data
    from targetdiff_amd import capi, models
    drv = reference_loader.load_driver()
    sd = weights.make_state_dict(SEED)
    cfg = dict(weights.DEFAULT_MODEL_CONFIG)
    mirror = models.ScorePosNet3D(cfg, weights.PROTEIN_FEATURE_DIM, weights.LIGAND_FEATURE_DIM)
    assert not mirror.load_state_dict(sd, strict=False).unexpected_keys
    log = []
    native = RecordingNative(sd, cfg, mirror.num_classes, log)
    monkeypatch.setattr(models.ScorePosNet3D, '_native', lambda self, device: native)
    monkeypatch.setattr(capi, 'NativeSession', StubSession)

    # the sampler refills fixed draw buffers in place (normal_ / uniform_); route the draws through randn_like / rand, which the
    # counter draws below patch, in the same order
    def draw(self, s):
        self._noise.copy_(torch.randn_like(self.lpos))
        if not self.pos_only:
            self._uniform.copy_(torch.rand(self.Nl, self.C, dtype=torch.float32))
    monkeypatch.setattr(models.ReverseSampler, '_draw', draw)
    # the mirror draws its per-step uniforms with torch.rand(N_l, K) where the reference calls torch.rand_like (inside
    # log_sample_categorical): route it through rand_like so that the counter draws patched over randn_like / rand_like serve both
    monkeypatch.setattr(torch, 'rand', lambda *size, **kw: torch.rand_like(torch.empty(*size, dtype=kw.get('dtype', torch.float32))))
    g = load_golden('driver_small.npz')
    steps = int(g['steps'])
    np.random.seed(SEED)
    with counter_draws(3100, lambda kind, n: n):
        res = drv.sample_diffusion_ligand(mirror, driver_data(), 5, batch_size=2, device='cpu', num_steps=steps, pos_only=False,
                                          center_pos_mode='protein', sample_num_atoms='prior')
    pos, v, pos_traj, v_traj, v0_traj, vt_traj, time_list = res
    # the call sequence the reference drove: three sample batches (2 + 2 + 1), each one session + `steps` x (forward, posterior)
    names = [n for n, _ in log]
    assert names.count('session_create') == 3 and names.count('model_forward') == 3 * steps == names.count('posterior_step')
    first_fwd = next(a for n, a in log if n == 'model_forward')
    assert first_fwd['protein_v'][1] == 'torch.float32' and first_fwd['ligand_v'][1] == 'torch.int64'
    assert len(time_list) == 3 and len(pos) == 5
    # ... and its outputs equal what the same driver produced with the reference's own model
    sizes = g['pos_n']
    assert [p.shape[0] for p in pos] == sizes.tolist() and pos[0].dtype == np.float64
    assert np.abs(np.concatenate(pos) - g['pos_cat']).max() < 5e-5
    assert np.array_equal(np.concatenate(v), g['v_cat'].astype(np.int64))
    assert pos_traj[0].shape == (steps, sizes[0], 3)
    assert np.abs(np.concatenate(pos_traj, axis=1) - g['pos_traj_cat']).max() < 5e-5
    assert np.array_equal(np.concatenate(v_traj, axis=1), g['v_traj_cat'].astype(np.int64))
    assert np.abs(np.concatenate(v0_traj, axis=1) - g['v0_traj_cat']).max() < 2e-4
    # pos_only / sample_num_atoms='ref' branch (:66-67, :108-112)
    log.clear()
    with counter_draws(3200, lambda kind, n: n):
        res2 = drv.sample_diffusion_ligand(mirror, driver_data(ref_ligand_atoms=9), 3, batch_size=2, device='cpu', num_steps=3,
                                           pos_only=True, center_pos_mode='protein', sample_num_atoms='ref')
    assert res2[4] == [] and res2[5] == []
    assert np.abs(np.concatenate(res2[0]) - g['po_pos_cat']).max() < 5e-5
    assert np.array_equal(np.concatenate(res2[1]), g['po_v_cat'].astype(np.int64))
