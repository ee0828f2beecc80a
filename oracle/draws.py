"""Portable, bit-reproducible pseudo-random draws for long parity runs (TEST INFRASTRUCTURE).

The reference draws its per-step noise with ``torch.randn_like`` / ``torch.rand_like``
(models/molopt_score_model.py:161,677).  Recording those draws for a 1000-step run would put megabytes of
incompressible floats into ``tests/golden``; regenerating them from a torch / numpy generator would tie the fixtures to
one library version.  Parity only needs *the same* numbers on both sides, so the long fixtures are produced by running
the real reference with ``randn_like`` / ``rand_like`` patched to the counter-based draws below, and the GPU tests
inject the very same arrays into the HIP sampler:

* integer hashing only (splitmix64 on ``(stream, step, element)`` counters, numpy uint64) -- no libm call, no
  library-specific generator, so every platform produces identical bits;
* ``uniform``  = top 24 bits / 2**24, exactly representable in fp32, in [0, 1);
* ``normal``   = Irwin-Hall: sum of 12 such uniforms - 6 (mean 0, variance 1, support [-6, 6]); sums of multiples of
  2**-24 are exact in float64, so the fp32 result is exact too.
"""
from __future__ import annotations

import numpy as np
import torch

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x: np.ndarray) -> np.ndarray:
    with np.errstate(over='ignore'):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        return z ^ (z >> np.uint64(31))


def _bits(stream: int, step: int, n: int, lanes: int = 1) -> np.ndarray:
    """[n, lanes] uint64 hashes of the counters (stream, step, element, lane)."""
    with np.errstate(over='ignore'):
        base = _splitmix64(np.asarray([(stream * 1000003 + step * 7919 + 12345) & 0xFFFFFFFFFFFFFFFF], dtype=np.uint64))
        idx = np.arange(n * lanes, dtype=np.uint64).reshape(n, lanes)
        return _splitmix64(base + idx * np.uint64(0x9E3779B97F4A7C15))


def uniform(stream: int, step: int, shape) -> torch.Tensor:
    n = int(np.prod(shape))
    u = (_bits(stream, step, n)[:, 0] >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy(u.astype(np.float32).reshape(shape))


def normal(stream: int, step: int, shape) -> torch.Tensor:
    n = int(np.prod(shape))
    u = (_bits(stream + 0x5bd1, step, n, 12) >> np.uint64(40)).astype(np.float64) / float(1 << 24)
    return torch.from_numpy((u.sum(axis=1) - 6.0).astype(np.float32).reshape(shape))


class Source:
    """``noise_source(step, name, like)`` callable for targetdiff_amd's sampler (and the patch target for the
    reference's ``randn_like`` / ``rand_like``): stream ``base`` for Gaussian draws, ``base + 1`` for uniforms."""

    def __init__(self, base: int, device=None):
        self.base, self.device = int(base), device

    def noise(self, step: int, shape):
        t = normal(self.base, step, tuple(shape))
        return t.to(self.device) if self.device is not None else t

    def uniform(self, step: int, shape):
        t = uniform(self.base + 1, step, tuple(shape))
        return t.to(self.device) if self.device is not None else t

    def __call__(self, step, name, like):
        return self.noise(step, like.shape) if name == 'noise' else self.uniform(step, like.shape)
