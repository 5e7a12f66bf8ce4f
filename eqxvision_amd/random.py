"""Stand-in for `jax.random` keys.

The reference threads `jax.random.PRNGKey`s through every `__call__` (e.g. resnet.py:343).  In inference they are dead values
(Dropout / DropPath are identities); in TRAINING mode they decide which activations are dropped (drop_path.py:37-61,
eqx.nn.Dropout), so the key algebra is JAX's: a key is `uint32[2]`, `split` and the random bits are Threefry-2x32 (20 rounds,
the Random123 generator `jax.random` uses) in JAX's counter layout (`threefry_partitionable` off, the default of the jax
versions equinox 0.9 ran on): for n values the counters 0 .. n-1 (one zero appended when n is odd) are cut in two halves that
form the (x0, x1) pairs; the outputs are concatenated [all x0', all x1'] and truncated to n.  Pinned by the Random123
known-answer vectors and by `split(PRNGKey(0))` / `uniform(PRNGKey(0))` as printed in JAX's documentation
(tests/test_host.py).  `bernoulli(key, p, shape)` = `uniform(key, shape) < p`, uniform = bits >> 9 | 0x3f800000 as float - 1.

Batched keys `[B, 2]` (what the caller hands to `vmap(net)(x, key=keys)`): `split` returns `[num, B, 2]` -- child i of
every sample -- and `bernoulli` one row of draws per sample, which is what `jax.vmap` over the key axis computes.

Parameter INITIALISATION (`uniform` / `normal` / `truncated_normal` below) still draws from numpy's PCG64 seeded with the
key: initial weights differ from the reference's for the same seed; pretrained / explicit weights are unaffected.
"""
from __future__ import annotations

import numpy as np


def PRNGKey(seed: int) -> np.ndarray:
    seed = int(seed)
    return np.array([(seed >> 32) & 0xFFFFFFFF, seed & 0xFFFFFFFF], dtype=np.uint32)


def _rotl(x, r):
    return (x << np.uint32(r)) | (x >> np.uint32(32 - r))


def threefry2x32(k0, k1, x0, x1):
    """Threefry-2x32, 20 rounds (Salmon et al., SC'11; jax/_src/prng.py `threefry2x32`): keys and counters broadcast."""
    k0, k1 = np.asarray(k0, np.uint32), np.asarray(k1, np.uint32)
    x0, x1 = np.asarray(x0, np.uint32), np.asarray(x1, np.uint32)
    ks = (k0, k1, k0 ^ k1 ^ np.uint32(0x1BD11BDA))
    rot = ((13, 15, 26, 6), (17, 29, 16, 24))
    with np.errstate(over="ignore"):
        x0 = x0 + ks[0]
        x1 = x1 + ks[1]
        for i in range(5):
            for r in rot[i % 2]:
                x0 = x0 + x1
                x1 = _rotl(x1, r) ^ x0
            x0 = x0 + ks[(i + 1) % 3]
            x1 = x1 + ks[(i + 2) % 3] + np.uint32(i + 1)
    return x0.astype(np.uint32), x1.astype(np.uint32)


def random_bits(key, n: int) -> np.ndarray:
    """n uint32 words for one key `[2]` -> `[n]`, or for batched keys `[B, 2]` -> `[B, n]` (jax `_threefry_random_bits`)."""
    key = np.asarray(key, np.uint32)
    n = int(n)
    c = np.arange(n + (n & 1), dtype=np.uint32)
    if n & 1:
        c[-1] = 0                                  # the padding counter
    h = c.size // 2
    if key.ndim == 2 and key.shape[0] > h:
        # many keys, few words each (per-token key splits): counters on the LEADING axis, so that numpy's inner loops run
        # over the keys -- [B, 2]-shaped operands cost 100x the time of [2, B]-shaped ones
        o0, o1 = threefry2x32(key[:, 0][None, :], key[:, 1][None, :], c[:h, None], c[h:, None])
        return np.ascontiguousarray(np.concatenate([o0, o1], axis=0)[:n].T)
    k0, k1 = key[..., 0:1], key[..., 1:2]
    o0, o1 = threefry2x32(k0, k1, c[:h], c[h:])
    return np.concatenate([o0, o1], axis=-1)[..., :n]


def split(key, num: int = 2) -> np.ndarray:
    """`jax.random.split`: `[2]` -> `[num, 2]`; batched `[B, 2]` -> `[num, B, 2]` (child i of every sample)."""
    key = np.asarray(key, np.uint32)
    num = int(num)
    b = random_bits(key, 2 * num)
    if key.ndim == 1:
        return b.reshape(num, 2)
    return np.ascontiguousarray(b.reshape(key.shape[0], num, 2).transpose(1, 0, 2))


def uniform01(key, n: int) -> np.ndarray:
    """`jax.random.uniform(key, (n,))` in [0, 1): mantissa bits of the random words."""
    b = random_bits(key, n)
    return ((b >> np.uint32(9)) | np.uint32(0x3F800000)).view(np.float32) - np.float32(1.0)


def bernoulli(key, p: float, shape=()) -> np.ndarray:
    """`jax.random.bernoulli(key, p, shape)`: bool `shape` for one key, `[B, *shape]` for batched keys."""
    key = np.asarray(key, np.uint32)
    n = int(np.prod(shape)) if len(tuple(shape)) else 1
    m = uniform01(key, n) < np.float32(p)
    lead = () if key.ndim == 1 else (key.shape[0],)
    return m.reshape(lead + tuple(shape))


def _seedseq(key) -> np.random.SeedSequence:
    k = np.asarray(key, dtype=np.uint32).reshape(-1)
    return np.random.SeedSequence([int(v) for v in k])


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
