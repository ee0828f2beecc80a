"""One process per GPU on one node, without an external launcher.

The reference's multi-GPU recipe is "run N shells" (README.md:96-102: ``CUDA_VISIBLE_DEVICES=k bash
scripts/batch_sample_diffusion.sh <cfg> <out> N k 0``).  :func:`spawn_ranks` is that recipe as a function: it starts N
copies of a command with the ``torch.distributed`` environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR /
MASTER_PORT) set exactly as ``python -m torch.distributed.run --nnodes=1 --nproc-per-node N`` would, so a script that
reads that environment behaves the same under either launcher.  By default all GPUs stay visible to every rank and a rank
selects its device with ``torch.cuda.set_device(LOCAL_RANK)`` (the torchrun convention); ``pin_devices=True`` (or
``TD_PIN_DEVICES=1``) is the reference's own recipe instead: rank r sees GPU r only (``HIP_VISIBLE_DEVICES=r``,
LOCAL_RANK = 0), so nothing a rank allocates can land on another rank's device.
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


def visible_device_list(env=None):
    """The outer restriction on the devices this job may use: the entries of HIP_VISIBLE_DEVICES, else of
    CUDA_VISIBLE_DEVICES (ROCm honours both); None when neither is set."""
    env = os.environ if env is None else env
    for name in ('HIP_VISIBLE_DEVICES', 'CUDA_VISIBLE_DEVICES'):
        if env.get(name, '') != '':
            return [d for d in env[name].split(',') if d != '']
    return None


def rank_env(rank: int, world: int, port: int, base=None, pin_devices: bool = False) -> dict:
    env = dict(os.environ if base is None else base)
    env.update(RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world), LOCAL_WORLD_SIZE=str(world),
               MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    if pin_devices:
        # the reference's recipe (README.md:96-102, CUDA_VISIBLE_DEVICES=k per shell): rank r sees one GPU, as device 0.
        # An outer device list is honoured -- HIP_VISIBLE_DEVICES, else CUDA_VISIBLE_DEVICES (a scheduler may hand out devices
        # through either): rank r gets its r-th entry, and a job with more ranks than entries is refused rather than placed on
        # GPUs outside the set it was restricted to.
        outer = visible_device_list(env)
        if outer is not None and rank >= len(outer):
            raise RuntimeError(f'pin_devices: rank {rank} of {world} has no device in the outer visible-device list {outer}')
        env['HIP_VISIBLE_DEVICES'] = outer[rank] if outer is not None else str(rank)
        env.pop('CUDA_VISIBLE_DEVICES', None)
        env['LOCAL_RANK'] = '0'
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # dmabuf IPC: RCCL between processes needs it on this driver
    return env


def _stop(procs, grace: float = 3.0):
    """terminate, then kill, every rank that is still running (their process groups: a rank may have children)"""
    import signal
    import time
    live = [p for p in procs if p.poll() is None]
    for p in live:
        try:
            os.killpg(p.pid, signal.SIGTERM)
        except (ProcessLookupError, PermissionError):
            p.terminate()
    t_end = time.monotonic() + grace
    while any(p.poll() is None for p in live) and time.monotonic() < t_end:
        time.sleep(0.05)
    for p in live:
        if p.poll() is None:
            try:
                os.killpg(p.pid, signal.SIGKILL)
            except (ProcessLookupError, PermissionError):
                p.kill()


class _Terminated(BaseException):
    """raised inside spawn_ranks by its SIGTERM / SIGHUP handler, so that the `finally` that stops the ranks runs"""

    def __init__(self, signum):
        super().__init__(signum)
        self.signum = signum


_ADDR_IN_USE = ('EADDRINUSE', 'Address already in use', 'address already in use', 'errno: 98')


def spawn_ranks(argv, nprocs: int, timeout=None, extra_env=None, pin_devices=None, _retries: int = 1) -> int:
    """Run ``argv`` (a full command line, e.g. ``[sys.executable, 'bench.py', ...]``) as ``nprocs`` ranks.  Rank 0
    inherits stdout (so its single JSON line is the job's output); every rank's stderr is passed through (and its tail kept,
    see below).  Returns the first non-zero exit code (0 if all ranks succeeded; 124 on a timeout; 128 + signal when this
    process was told to stop).  Whatever ends the supervision -- a failing rank, the timeout, KeyboardInterrupt, SIGTERM /
    SIGHUP delivered to this process (handlers are installed for the duration of the call when it runs in the main thread:
    Python's default action for SIGTERM would end the supervisor without running any clean-up, and the ranks, being their own
    sessions, are out of reach of a process-group kill aimed at the supervisor), an exception while starting rank k > 0 --
    every rank still running is terminated (each rank is its own session, so its children go with it): no orphan keeps a GPU
    or the rendezvous port.  The port is picked free at start; when a rank fails within the first seconds AND its stderr shows
    the rendezvous could not bind the port (EADDRINUSE: another job took it in between), the job is retried once on a new port."""
    import collections
    import signal
    import threading
    import time
    if pin_devices is None:
        pin_devices = os.environ.get('TD_PIN_DEVICES', '0') not in ('', '0')
    port = free_port()
    procs = []
    rc = 0
    timed_out = False
    stopped_by = None
    t_start = time.monotonic()
    tails = collections.deque(maxlen=400)          # last stderr lines of all ranks (looked at only to classify a failed start)
    pumps = []

    def _pump(stream):
        for line in iter(stream.readline, b''):
            try:
                sys.stderr.buffer.write(line)
                sys.stderr.buffer.flush()
            except Exception:
                pass
            tails.append(line.decode('utf-8', 'replace'))
        stream.close()

    cleaning_up = []        # non-empty once the `finally` below has started: a second signal is noted, not raised

    def _on_signal(signum, frame):
        if cleaning_up:
            cleaning_up.append(signum)
            return
        raise _Terminated(signum)

    old_handlers = {}
    if threading.current_thread() is threading.main_thread():
        for sig in (signal.SIGTERM, signal.SIGHUP):
            try:
                old_handlers[sig] = signal.signal(sig, _on_signal)
            except (ValueError, OSError):
                pass
    try:
        for r in range(nprocs):
            env = rank_env(r, nprocs, port, pin_devices=pin_devices)
            if extra_env:
                env.update(extra_env)
            p = subprocess.Popen(list(argv), env=env, stdout=None if r == 0 else subprocess.DEVNULL, stderr=subprocess.PIPE,
                                 start_new_session=True)
            procs.append(p)
            t = threading.Thread(target=_pump, args=(p.stderr,), daemon=True)
            t.start()
            pumps.append(t)
        deadline = None if timeout is None else t_start + timeout
        live = list(procs)
        while live:                   # poll all ranks: one that dies must not leave the others waiting in a rendezvous
            for p in list(live):
                code = p.poll()
                if code is None:
                    continue
                live.remove(p)
                if code != 0 and rc == 0:
                    rc = code
                    _stop(live)
            if live and deadline is not None and time.monotonic() > deadline:
                timed_out = True
                _stop(live)
            if live:
                time.sleep(0.05)
    except _Terminated as t:
        stopped_by = t.signum
    finally:
        cleaning_up.append(0)          # from here on SIGTERM / SIGHUP must not interrupt the clean-up (the handlers only take note)
        _stop(procs)
        for sig, h in old_handlers.items():
            try:
                signal.signal(sig, h)
            except (ValueError, OSError):
                pass
        for t in pumps:
            t.join(timeout=2.0)
    if stopped_by is None and len(cleaning_up) > 1:
        stopped_by = cleaning_up[1]          # told to stop while cleaning up
    if stopped_by is not None:
        return 128 + int(stopped_by)
    if timed_out:
        return 124
    lines = list(tails)          # a snapshot: a pump thread that outlived its join (a grandchild holding stderr open) may still append
    if rc != 0 and _retries > 0 and time.monotonic() - t_start < 20.0 and any(k in line for line in lines for k in _ADDR_IN_USE):
        return spawn_ranks(argv, nprocs, timeout, extra_env, pin_devices, _retries - 1)
    return rc


def _port_in_use(port: int) -> bool:
    """after a failed start: is someone (else) listening on the rendezvous port?"""
    with socket.socket() as s:
        s.settimeout(0.2)
        return s.connect_ex(('127.0.0.1', port)) == 0


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


def _device_identity(device) -> dict:
    """What tells two ranks' devices apart: ordinal, PCI address and UUID of a HIP device (whatever torch exposes); for a CPU
    'device' only its name."""
    import torch
    device = torch.device(device)
    ident = {'device': str(device), 'host': socket.gethostname(), 'pid': os.getpid()}
    if device.type == 'cuda':
        idx = device.index if device.index is not None else torch.cuda.current_device()
        props = torch.cuda.get_device_properties(idx)
        ident.update(ordinal=int(idx), name=props.name, visible_devices=os.environ.get('HIP_VISIBLE_DEVICES', os.environ.get('CUDA_VISIBLE_DEVICES')))
        bus = [getattr(props, k, None) for k in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')]
        ident['pci'] = ':'.join('%04x' % bus[0] if i == 0 else '%02x' % b for i, b in enumerate(bus)) if None not in bus else None
        uuid = getattr(props, 'uuid', None)
        ident['uuid'] = str(uuid) if uuid is not None else None
    return ident


def rank_census(device, seconds_per_step=None, extra=None) -> dict:
    """What the process group actually looks like, gathered from every rank (one object gather: metadata, not data path): world
    size and backend as torch.distributed reports them, and per rank its RANK / LOCAL_RANK, device ordinal, PCI address, UUID, host,
    pid and -- when given -- its own seconds per step.  Two ranks on the same physical device (same host and PCI address or UUID)
    are an error: the job would report N GPUs while running on fewer.  Works without a process group (a census of one)."""
    import torch.distributed as dist
    me = _device_identity(device)
    me.update(rank=int(os.environ.get('RANK', '0')), local_rank=int(os.environ.get('LOCAL_RANK', '0')))
    if seconds_per_step is not None:
        me['ms_per_step'] = float(seconds_per_step) * 1e3
    if extra:
        me.update(extra)
    if dist.is_available() and dist.is_initialized():
        world, backend = dist.get_world_size(), dist.get_backend()
        me['rank'] = dist.get_rank()
        ranks = [None] * world
        dist.all_gather_object(ranks, me)
    else:
        world, backend, ranks = 1, None, [me]
    ranks = sorted(ranks, key=lambda r: r['rank'])
    seen = {}
    for r in ranks:
        key = (r['host'], r.get('uuid') or r.get('pci'))
        if r.get('ordinal') is not None and key[1] is not None:
            if key in seen:
                raise RuntimeError(f"ranks {seen[key]} and {r['rank']} run on the same device {key[1]} of host {key[0]}: "
                                   'the job would report more GPUs than it uses (check LOCAL_RANK / HIP_VISIBLE_DEVICES)')
            seen[key] = r['rank']
    out = {'world_size': int(world), 'backend': backend, 'env_world_size': int(os.environ.get('WORLD_SIZE', '1')), 'ranks': ranks}
    if all('ms_per_step' in r for r in ranks):
        ms = [r['ms_per_step'] for r in ranks]
        out['ms_per_step_per_rank'] = ms
        out['max_over_mean'] = max(ms) / (sum(ms) / len(ms))
    return out


class stdout_to_stderr:
    """File descriptor 1 points at stderr while the block runs.  RCCL prints a version banner ("RCCL version : ...", five lines) to
    STDOUT when its first communicator is created; rank 0's stdout is the job's one JSON line, so the process group is set up inside
    this block (init_process_group + a first barrier, which forces the communicator into existence)."""

    def __enter__(self):
        sys.stdout.flush()
        self._saved = os.dup(1)
        os.dup2(2, 1)
        return self

    def __exit__(self, *exc):
        sys.stdout.flush()
        os.dup2(self._saved, 1)
        os.close(self._saved)
        return False
