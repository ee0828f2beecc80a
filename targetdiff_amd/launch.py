"""One process per GPU on one node, without an external launcher.

The reference's multi-GPU recipe is "run N shells" (README.md:96-102: ``CUDA_VISIBLE_DEVICES=k bash
scripts/batch_sample_diffusion.sh <cfg> <out> N k 0``).  :func:`spawn_ranks` is that recipe as a function: it starts N
copies of a command with the ``torch.distributed`` environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT) set exactly as ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would, so a script that
reads that environment behaves the same under either launcher.  All GPUs stay visible to every rank; a rank selects its
device with ``torch.cuda.set_device(LOCAL_RANK)`` (the torchrun convention).
"""
from __future__ import annotations

import os
import socket
import subprocess
import sys


def free_port() -> int:
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        return int(s.getsockname()[1])


def under_launcher() -> bool:
    """True when the process already runs as one rank of a launched job (torchrun or spawn_ranks)."""
    return 'RANK' in os.environ and 'WORLD_SIZE' in os.environ


def rank_env(rank: int, world: int, port: int, base=None) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this driver
    return env


def spawn_ranks(argv, nprocs: int, timeout=None, extra_env=None) -> int:
    """Run ``argv`` (a full command line, e.g. ``[sys.executable, 'bench.py', ...]``) as ``nprocs`` ranks.  Rank 0
    inherits stdout (so its single JSON line is the job's output); every rank inherits stderr.  Returns the first
    non-zero exit code (0 if all ranks succeeded); on a failure the remaining ranks are terminated."""
    port = free_port()
    procs = []
    for r in range(nprocs):
        env = rank_env(r, nprocs, port)
        if extra_env:
            env.update(extra_env)
        procs.append(subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL))
    import time
    rc = 0
    deadline = None if timeout is None else time.monotonic() + timeout
    live = list(procs)
    while live:                       # poll all ranks: one that dies must not leave the others waiting in a rendezvous
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0 and rc == 0:
                rc = code
                for q in live:
                    q.terminate()
        if live and deadline is not None and time.monotonic() > deadline:
            rc = rc or 124
            for q in live:
                q.kill()
        if live:
            time.sleep(0.05)
    return rc


def self_spawn_if_needed(n_gpus: int) -> bool:
    """``python script.py --gpus N`` with no launcher around it: re-run the same command line as N ranks and exit with
    their status.  Returns False when nothing had to be spawned (N == 1, or already inside a launched rank)."""
    if n_gpus <= 1 or under_launcher():
        return False
    try:
        import torch
        visible = torch.cuda.device_count()
    except Exception:
        visible = None
    if visible is not None and n_gpus > visible:
        raise SystemExit(f'--gpus {n_gpus}: only {visible} HIP device(s) visible to this process')
    sys.stdout.flush()
    raise SystemExit(spawn_ranks([sys.executable] + sys.argv, n_gpus))
