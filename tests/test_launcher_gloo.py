"""The multi-GPU launch path on CPU: targetdiff_amd.launch.spawn_ranks starts 2 ranks (the way `bench.py --gpus N` and
`tools/batch_sample.py --gpus N` start N), each rank runs tools/batch_sample.py's main with the gloo backend and a
stand-in model, writes result_{i}.pt files in the reference layout and joins the metadata gather."""
import json
import os
import sys

import numpy as np
import torch

from targetdiff_amd import launch, results

HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, extra=()):
    argv = [sys.executable, os.path.join(HERE, '_gloo_batch_worker.py'), '--pockets', 'synthetic:5', '--result_path',
            str(tmp_path), '--num_samples', '3', '--num_steps', '2', '--batch_size', '2', '--ligand_atoms', '4',
            '--device', 'cpu', *extra]
    rc = launch.spawn_ranks(argv, 2, timeout=300)
    assert rc == 0
    with open(os.path.join(str(tmp_path), 'summary.json')) as f:
        return json.load(f)


def test_two_ranks_write_reference_layout_results_and_resume(tmp_path):
    s = _run(tmp_path)
    assert s['world_size'] == 2 and s['pockets_sampled'] == 5 and s['ligands'] == 15
    per_rank = {m['rank']: sorted(p['pocket'] for p in m['pockets']) for m in s['per_rank']}
    assert per_rank == {0: [0, 2, 4], 1: [1, 3]}                        # scripts/batch_sample_diffusion.sh:15-17
    for i in range(5):
        r = torch.load(results.result_file(str(tmp_path), i), weights_only=False)
        # the keys scripts/evaluate_diffusion.py:70-76 reads
        assert set(r) == {'data', 'pred_ligand_pos', 'pred_ligand_v', 'pred_ligand_pos_traj', 'pred_ligand_v_traj', 'time'}
        assert len(r['pred_ligand_pos']) == 3 and r['pred_ligand_pos'][0].shape == (4, 3)
        assert r['pred_ligand_pos'][0].dtype == np.float64
        assert r['pred_ligand_pos_traj'][0].shape == (2, 4, 3) and r['pred_ligand_v_traj'][0].shape == (2, 4)
        assert len(r['time']) == 2                                       # two sample batches (2 + 1)
    assert not [f for f in os.listdir(str(tmp_path)) if '.tmp.' in f]
    # a re-run finds every file and samples nothing; after removing one file only that pocket is sampled again
    s2 = _run(tmp_path)
    assert s2['pockets_sampled'] == 0 and s2['pockets_skipped'] == 5
    os.remove(results.result_file(str(tmp_path), 3))
    s3 = _run(tmp_path)
    assert s3['pockets_sampled'] == 1 and s3['pockets_skipped'] == 4
    # START_IDX (scripts/batch_sample_diffusion.sh:13,15)
    os.remove(results.result_file(str(tmp_path), 0))
    s4 = _run(tmp_path, ('--start_idx', '2'))
    assert s4['pockets_sampled'] == 0 and s4['pockets_skipped'] == 3
    assert not os.path.exists(results.result_file(str(tmp_path), 0))


def test_rank_environment_matches_torchrun():
    env = launch.rank_env(3, 8, 29511, base={})
    assert env['RANK'] == '3' and env['LOCAL_RANK'] == '3' and env['WORLD_SIZE'] == '8'
    assert env['MASTER_ADDR'] == '127.0.0.1' and env['MASTER_PORT'] == '29511'
    assert env['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'


def test_a_failing_rank_ends_the_job(tmp_path):
    """One rank exits with an error while the other would wait forever: spawn_ranks must notice, stop the rest and report."""
    import time
    script = tmp_path / 'w.py'
    script.write_text('import os, sys, time\n'
                      'if os.environ["RANK"] == "1":\n    sys.exit(3)\n'
                      'time.sleep(600)\n')
    t0 = time.time()
    rc = launch.spawn_ranks([sys.executable, str(script)], 2, timeout=120)
    assert rc == 3 and time.time() - t0 < 30


def test_self_spawn_refuses_more_ranks_than_devices(monkeypatch):
    import pytest
    import torch
    monkeypatch.delenv('RANK', raising=False)
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    assert launch.self_spawn_if_needed(1) is False
    with pytest.raises(SystemExit, match='visible'):
        launch.self_spawn_if_needed(torch.cuda.device_count() + 2)


def test_pinned_rank_environment():
    """pin_devices: the reference's CUDA_VISIBLE_DEVICES recipe (README.md:96-102) -- one visible GPU per rank, as device 0."""
    from targetdiff_amd import launch
    env = launch.rank_env(3, 8, 29501, base={}, pin_devices=True)
    assert env['HIP_VISIBLE_DEVICES'] == '3' and env['LOCAL_RANK'] == '0' and env['RANK'] == '3' and env['WORLD_SIZE'] == '8'
    env = launch.rank_env(1, 2, 29501, base={'HIP_VISIBLE_DEVICES': '4,6'}, pin_devices=True)
    assert env['HIP_VISIBLE_DEVICES'] == '6'
    assert launch.rank_env(3, 8, 29501, base={})['LOCAL_RANK'] == '3'


def test_timeout_and_interrupt_leave_no_rank_behind(tmp_path):
    """A hung job: spawn_ranks returns 124 and every rank (and its child) is gone; the same clean-up runs when the
    supervising loop is interrupted."""
    import subprocess
    import sys
    import time
    from targetdiff_amd import launch
    script = tmp_path / 'hang.py'
    script.write_text('import os, subprocess, sys, time\n'
                      'open(os.path.join(sys.argv[1], "pid_%s" % os.environ["RANK"]), "w").write(str(os.getpid()))\n'
                      'c = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])\n'
                      'open(os.path.join(sys.argv[1], "child_%s" % os.environ["RANK"]), "w").write(str(c.pid))\n'
                      'time.sleep(600)\n')
    t0 = time.time()
    rc = launch.spawn_ranks([sys.executable, str(script), str(tmp_path)], 2, timeout=3.0)
    assert rc == 124 and time.time() - t0 < 30
    pids = [int((tmp_path / f).read_text()) for f in ('pid_0', 'pid_1', 'child_0', 'child_1')]
    time.sleep(0.5)
    for pid in pids:
        alive = subprocess.run(['ps', '-p', str(pid), '-o', 'stat='], capture_output=True, text=True).stdout.strip()
        assert alive == '' or alive.startswith('Z'), (pid, alive)


def test_sigterm_to_the_supervisor_leaves_no_rank_behind(tmp_path):
    """SIGTERM (what `timeout`, a CI runner or a scheduler sends) to the process that supervises the ranks: Python's default action
    would end it without any clean-up and the ranks -- their own sessions -- would keep their GPUs and the rendezvous port.
    spawn_ranks installs a handler for the duration of the call: every rank and its child is gone, exit status 128 + SIGTERM."""
    import signal
    import subprocess
    import sys
    import time
    hang = tmp_path / 'hang.py'
    hang.write_text('import os, subprocess, sys, time\n'
                    'c = subprocess.Popen([sys.executable, "-c", "import time; time.sleep(600)"])\n'
                    'open(os.path.join(sys.argv[1], "child_%s" % os.environ["RANK"]), "w").write(str(c.pid))\n'
                    'open(os.path.join(sys.argv[1], "pid_%s" % os.environ["RANK"]), "w").write(str(os.getpid()))\n'
                    'time.sleep(600)\n')
    sup = tmp_path / 'sup.py'
    sup.write_text('import sys\n'
                   f'sys.path.insert(0, {os.path.dirname(HERE)!r})\n'
                   'from targetdiff_amd import launch\n'
                   'sys.exit(launch.spawn_ranks([sys.executable, sys.argv[1], sys.argv[2]], 2, timeout=300))\n')
    p = subprocess.Popen([sys.executable, str(sup), str(hang), str(tmp_path)])
    t_end = time.time() + 60
    names = ('pid_0', 'pid_1', 'child_0', 'child_1')
    while time.time() < t_end and not all((tmp_path / f).exists() and (tmp_path / f).read_text() for f in names):
        time.sleep(0.1)
    pids = [int((tmp_path / f).read_text()) for f in names]
    p.send_signal(signal.SIGTERM)
    rc = p.wait(timeout=30)
    assert rc == 128 + signal.SIGTERM
    time.sleep(0.5)
    for pid in pids:
        alive = subprocess.run(['ps', '-p', str(pid), '-o', 'stat='], capture_output=True, text=True).stdout.strip()
        assert alive == '' or alive.startswith('Z'), (pid, alive)
    # the handlers are gone again in a process that returned from spawn_ranks normally
    import signal as _s
    before = _s.getsignal(_s.SIGTERM)
    ok = tmp_path / 'ok.py'
    ok.write_text('pass\n')
    assert launch.spawn_ranks([sys.executable, str(ok)], 2, timeout=60) == 0
    assert _s.getsignal(_s.SIGTERM) is before


def test_pinned_devices_honour_the_outer_restriction():
    """pin_devices under a scheduler that hands out devices through CUDA_VISIBLE_DEVICES (ROCm honours it too): rank r gets the
    r-th entry of the outer list, and a job with more ranks than entries is refused instead of spilling onto other GPUs."""
    import pytest
    env = launch.rank_env(1, 2, 29501, base={'CUDA_VISIBLE_DEVICES': '5,7'}, pin_devices=True)
    assert env['HIP_VISIBLE_DEVICES'] == '7' and 'CUDA_VISIBLE_DEVICES' not in env and env['LOCAL_RANK'] == '0'
    env = launch.rank_env(0, 2, 29501, base={'HIP_VISIBLE_DEVICES': '2,3', 'CUDA_VISIBLE_DEVICES': '5,7'}, pin_devices=True)
    assert env['HIP_VISIBLE_DEVICES'] == '2'                       # HIP_VISIBLE_DEVICES wins
    with pytest.raises(RuntimeError, match='no device'):
        launch.rank_env(2, 3, 29501, base={'HIP_VISIBLE_DEVICES': '4,6'}, pin_devices=True)
    with pytest.raises(RuntimeError, match='no device'):
        launch.rank_env(1, 2, 29501, base={'CUDA_VISIBLE_DEVICES': '4'}, pin_devices=True)


def test_retry_only_when_the_rendezvous_port_was_taken(tmp_path):
    """A rank that fails at once is retried on a new port only when its stderr says the port could not be bound; any other
    early failure is reported as it is (one run), even if something unrelated listens on the port."""
    import sys
    count = tmp_path / 'runs'
    script = tmp_path / 'w.py'
    script.write_text('import os, sys\n'
                      f'open({str(count)!r}, "a").write(os.environ["RANK"] + "\\n")\n'
                      'if os.environ["RANK"] == "0":\n'
                      '    sys.stderr.write(sys.argv[1] + "\\n"); sys.exit(7)\n')
    assert launch.spawn_ranks([sys.executable, str(script), 'some other failure'], 2, timeout=60) == 7
    assert count.read_text().count('0') == 1
    count.write_text('')
    assert launch.spawn_ranks([sys.executable, str(script), 'RuntimeError: Address already in use (errno: 98)'], 2, timeout=60) == 7
    assert count.read_text().count('0') == 2                        # retried once
