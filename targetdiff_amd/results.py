"""``result_{data_id}.pt`` files in the reference's layout, written off the sampling thread.

The reference driver ends every pocket with a synchronous ``torch.save`` of the whole trajectory
(scripts/sample_diffusion.py:175-188; ~36 MB per pocket at 100 samples x 1000 steps), consumed
by scripts/evaluate_diffusion.py:70-76 (``r['pred_ligand_pos_traj']``, ``r['pred_ligand_v_traj']``).  Here the file is
serialised by a background thread while the GPU already samples the next pocket, and lands under its final name by an
atomic rename, so the skip-if-exists resume (the reference's manual ``START_IDX``, scripts/batch_sample_diffusion.sh:13)
never sees a partial file.
"""
from __future__ import annotations

import os
import queue
import threading

import torch


def result_file(result_path: str, data_id) -> str:
    return os.path.join(result_path, f'result_{data_id}.pt')          # scripts/sample_diffusion.py:188


def result_dict(data, sampled) -> dict:
    """The dictionary scripts/sample_diffusion.py:175-182 saves, from the driver's 7-tuple."""
    pred_pos, pred_v, pred_pos_traj, pred_v_traj, _v0, _vt, time_list = sampled
    return {'data': data, 'pred_ligand_pos': pred_pos, 'pred_ligand_v': pred_v, 'pred_ligand_pos_traj': pred_pos_traj,
            'pred_ligand_v_traj': pred_v_traj, 'time': time_list}


def save_result(path: str, obj: dict) -> None:
    os.makedirs(os.path.dirname(os.path.abspath(path)), exist_ok=True)
    tmp = f'{path}.tmp.{os.getpid()}'
    torch.save(obj, tmp)
    os.replace(tmp, path)


class AsyncResultWriter:
    """Single background thread; ``submit`` returns at once, ``close`` waits for the queue and re-raises a failure."""

    def __init__(self, max_pending: int = 2):
        self._q = queue.Queue(maxsize=max_pending)        # bounds host memory: at most `max_pending` pockets in flight
        self._err = None
        self._t = threading.Thread(target=self._run, daemon=True)
        self._t.start()
        self.written = []

    def _run(self):
        while True:
            item = self._q.get()
            if item is None:
                return
            path, obj = item
            try:
                save_result(path, obj)
                self.written.append(path)
            except BaseException as exc:      # surfaced by close() / the next submit()
                self._err = exc

    def submit(self, path: str, obj: dict) -> None:
        if self._err is not None:
            raise self._err
        self._q.put((path, obj))

    def close(self) -> None:
        self._q.put(None)
        self._t.join()
        if self._err is not None:
            raise self._err

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False
