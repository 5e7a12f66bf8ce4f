"""Minimal stand-in for `jax.random` keys.

The reference threads `jax.random.PRNGKey`s through every `__call__` (e.g. resnet.py:343) but in
inference they are dead values (Dropout/DropPath are identities).  The drop-in keeps the calling
convention -- `key=` is required where the reference requires it -- without depending on jax:
a key is a `uint32[2]` numpy array; `split` derives children with `numpy.random.SeedSequence`.
JAX's threefry bit-stream is NOT reproduced (parameter init therefore differs from the
reference's for the same seed; pretrained/explicit weights are unaffected).
"""
from __future__ import annotations

import numpy as np


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def _seedseq(key) -> np.random.SeedSequence:
    k = np.asarray(key, dtype=np.uint32).reshape(-1)
    return np.random.SeedSequence([int(v) for v in k])


def split(key, num: int = 2) -> np.ndarray:
    ss = _seedseq(key)
    return np.stack([c.generate_state(2).astype(np.uint32) for c in ss.spawn(int(num))])


def generator(key) -> np.random.Generator:
    if key is None:
        key = PRNGKey(0)
    return np.random.Generator(np.random.PCG64(_seedseq(key)))


def uniform(key, shape=(), minval=0.0, maxval=1.0) -> np.ndarray:
    return generator(key).uniform(minval, maxval, size=shape).astype(np.float32)


def normal(key, shape=()) -> np.ndarray:
    return generator(key).standard_normal(size=shape).astype(np.float32)


def truncated_normal(key, lower, upper, shape=()) -> np.ndarray:
    g = generator(key)
    if lower == upper:                      # swin.py:303-312 quirk: degenerate interval -> constant
        return np.full(shape, float(lower), np.float32)
    x = g.standard_normal(size=shape)
    bad = (x < lower) | (x > upper)
    while bad.any():
        x[bad] = g.standard_normal(size=int(bad.sum()))
        bad = (x < lower) | (x > upper)
    return x.astype(np.float32)
