"""Minimal `eqx.Module` work-alike: dataclass-style pytrees without jax.

The reference builds every layer/model as a frozen-dataclass `eqx.Module` whose
pytree flatten order is the class-annotation order, depth first.  That order is
the *contract* of `load_torch_weights` (reference `eqxvision/utils.py:172-199`),
so this base class reproduces exactly that: fields = annotations in MRO order,
`tree_flatten` walks them in declaration order, descending into Modules / lists /
tuples / dicts.  Array leaves are host `numpy` arrays (the fp32 master copy);
device-side copies (bf16, re-laid-out, BN-folded) are built lazily by
`eqxvision_amd.ops` and cached on the module instance.
"""
from __future__ import annotations

import copy
from typing import Any, Callable, List, Tuple

import numpy as np


class StateIndex:
    """Stand-in for `eqx.experimental.StateIndex`: a mutable slot holding BatchNorm running
    statistics outside the parameter leaves (reference utils.py:203-218)."""

    __slots__ = ("_value", "version", "_dev", "_dev_newer", "_last_update")

    def __init__(self, value=None):
        self._value = value
        self.version = 0            # bumped by every update: weights prepared with the old statistics folded in are stale
        self._dev = None            # device copy of the statistics (training-mode steps update THIS, ops.bn_train_update)
        self._dev_newer = False     # ... and the host value is fetched when somebody asks for it
        self._last_update = None    # what the backward of the last training-mode call needs (ops.bn_train_update -> grad._bn_vectors)

    @property
    def value(self):
        if self._dev_newer:
            self._value = tuple(t.cpu().numpy() for t in self._dev)
            self._dev_newer = False
        return self._value

    @value.setter
    def value(self, v):
        self._value = v
        if self._dev is not None and v is not None:
            # a recorded training-mode step has the ADDRESSES of the device copy baked in (transforms: 'state-rw' signature):
            # refresh those tensors in place instead of dropping them, or the replay would keep updating the old statistics and
            # its hook would hand them back over the values set here (round-2 advice)
            import torch
            if len(v) != len(self._dev) or any(int(np.size(a)) != t.numel() for t, a in zip(self._dev, v)):
                # a recorded training step has the old tensors' ADDRESSES baked in: a value of another shape cannot be refreshed in
                # place, and silently dropping the device copy would leave that recording updating stale statistics
                raise ValueError("StateIndex.value: the new statistics do not match the shape of the device copy a recorded "
                                 "training step may hold; build a fresh module (or clear filter_jit's cache) instead")
            for t, a in zip(self._dev, v):
                t.copy_(torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32)).reshape(tuple(t.shape))))
        else:
            self._dev = None
        self._dev_newer = False
        self.version += 1

    def device_updated(self, tensors):
        """A training-mode step rewrote the statistics on the device: `tensors` are the current values."""
        self._dev, self._dev_newer = tuple(tensors), True
        self.version += 1

    def __copy__(self):
        new = StateIndex(self.value)
        new.version = self.version
        return new

    def __repr__(self):
        return f"StateIndex({'set' if self.value is not None else 'unset'})"


class _ModuleMeta(type):
    def __new__(mcls, name, bases, ns):
        if "__call__" in ns and callable(ns["__call__"]):
            from . import _trace
            ns = dict(ns)
            ns["__call__"] = _trace.wrap_call(name, ns["__call__"])    # roctx range per module call when EQV_ROCTX=1
        cls = super().__new__(mcls, name, bases, ns)
        fields: List[str] = []
        for klass in reversed(cls.__mro__):
            for f in klass.__dict__.get("__annotations__", {}):
                if f not in fields and not f.startswith("__"):
                    fields.append(f)
        cls.__fields__ = tuple(fields)
        return cls


class Module(metaclass=_ModuleMeta):
    """Base class.  Sub-classes declare fields as class annotations and assign them in
    `__init__`, like `eqx.Module`."""

    __fields__: Tuple[str, ...] = ()

    def __repr__(self):
        inner = ", ".join(f"{f}={_short(getattr(self, f, None))}" for f in self.__fields__)
        return f"{type(self).__name__}({inner})"

    # a per-instance cache for device-side prepared weights (NOT a pytree field)
    def _cache(self) -> dict:
        c = self.__dict__.get("_dev_cache")
        if c is None:
            c = {}
            object.__setattr__(self, "_dev_cache", c)
        return c


def _short(v):
    if isinstance(v, DevArray):
        return repr(v)
    if isinstance(v, np.ndarray):
        return f"f{v.dtype.itemsize * 8}{list(v.shape)}"
    if isinstance(v, (list, tuple)) and len(v) > 4:
        return f"[{len(v)} items]"
    return repr(v)


class DevArray:
    """An fp32 array leaf that LIVES ON THE DEVICE -- what a jax.Array leaf is in the reference: gradients, optimiser updates and the
    parameters of a model that has been through `apply_updates` (a training loop keeps its state in HBM; nothing crosses PCIe per
    step).  `shape` / `dtype` / `ndim` / `size` like numpy; `np.asarray(x)` -- which every host-side consumer goes through
    (state_dict, weight folding, printing) -- fetches a host copy ONCE and keeps it.  `.dev` is the device tensor."""
    __slots__ = ("dev", "_host", "shape", "__weakref__")
    __array_priority__ = 1000
    dtype = np.dtype(np.float32)

    def __init__(self, dev):
        self.dev = dev.detach()
        self.shape = tuple(int(d) for d in dev.shape)
        self._host = None

    @property
    def ndim(self):
        return len(self.shape)

    @property
    def size(self):
        return int(np.prod(self.shape)) if self.shape else 1

    def __len__(self):
        if not self.shape:
            raise TypeError("len() of a 0-d array")
        return self.shape[0]

    def _h(self):
        if self._host is None:
            self._host = self.dev.cpu().numpy()
            self._host.flags.writeable = False          # leaves are immutable, like the reference's arrays
        return self._host

    def __array__(self, dtype=None, copy=None):
        """numpy 2 protocol: `copy=True` must hand out memory the caller owns (np.array(leaf) is writable and does not alias the
        cached read-only host copy), `copy=False` must raise when a conversion would be needed, `copy=None` may share."""
        h = self._h()
        if dtype is not None and np.dtype(dtype) != h.dtype:
            if copy is False:
                raise ValueError("DevArray: a dtype conversion needs a copy (copy=False was requested)")
            return h.astype(dtype)
        return h.copy() if copy else h

    # numpy semantics on the host copy: ufuncs (np.sqrt(leaf), leaf ** 2 via the operators below), reductions and array functions
    # (np.linalg.norm, np.concatenate, np.clip ...) see a plain ndarray -- gradient clipping / L2 penalties written against the
    # reference's jax arrays keep working on device-resident leaves.
    def __array_ufunc__(self, ufunc, method, *inputs, **kwargs):
        un = lambda v: np.asarray(v) if isinstance(v, DevArray) else v
        if "out" in kwargs:
            if any(isinstance(o, DevArray) for o in kwargs["out"]):
                raise TypeError("DevArray leaves are immutable: cannot be a ufunc `out=` target")
        return getattr(ufunc, method)(*[un(v) for v in inputs], **kwargs)

    def __array_function__(self, func, types, args, kwargs):
        def un(v):
            if isinstance(v, DevArray):
                return np.asarray(v)
            if isinstance(v, (list, tuple)):
                return type(v)(un(e) for e in v)
            return v
        return func(*[un(a) for a in args], **{k: un(v) for k, v in kwargs.items()})

    def astype(self, dtype, copy=True):
        return np.asarray(self).astype(dtype, copy=copy)

    def reshape(self, *shape):
        return np.asarray(self).reshape(*shape)

    def copy(self):
        return np.asarray(self).copy()

    @property
    def T(self):
        return np.asarray(self).T

    def transpose(self, *axes): return np.asarray(self).transpose(*axes)                 # noqa: E704
    def sum(self, *a, **k): return np.asarray(self).sum(*a, **k)                         # noqa: E704
    def mean(self, *a, **k): return np.asarray(self).mean(*a, **k)                       # noqa: E704
    def max(self, *a, **k): return np.asarray(self).max(*a, **k)                         # noqa: E704
    def min(self, *a, **k): return np.asarray(self).min(*a, **k)                         # noqa: E704
    def std(self, *a, **k): return np.asarray(self).std(*a, **k)                         # noqa: E704
    def var(self, *a, **k): return np.asarray(self).var(*a, **k)                         # noqa: E704
    def ravel(self): return np.asarray(self).ravel()                                     # noqa: E704
    def flatten(self): return np.asarray(self).flatten()                                 # noqa: E704
    def item(self): return np.asarray(self).item()                                       # noqa: E704
    def tolist(self): return np.asarray(self).tolist()                                   # noqa: E704
    def __float__(self): return float(np.asarray(self))                                  # noqa: E704

    def __getitem__(self, i):
        return np.asarray(self)[i]

    def __iter__(self):
        return iter(np.asarray(self))

    def __repr__(self):
        return f"DevArray(f32{list(self.shape)}, on device)"

    def _np(self, other):
        return np.asarray(other) if isinstance(other, DevArray) else other

    def __add__(self, o): return np.asarray(self) + self._np(o)          # noqa: E704
    def __radd__(self, o): return self._np(o) + np.asarray(self)         # noqa: E704
    def __sub__(self, o): return np.asarray(self) - self._np(o)          # noqa: E704
    def __rsub__(self, o): return self._np(o) - np.asarray(self)         # noqa: E704
    def __mul__(self, o): return np.asarray(self) * self._np(o)          # noqa: E704
    def __rmul__(self, o): return self._np(o) * np.asarray(self)         # noqa: E704
    def __truediv__(self, o): return np.asarray(self) / self._np(o)      # noqa: E704
    def __rtruediv__(self, o): return self._np(o) / np.asarray(self)     # noqa: E704
    def __pow__(self, o): return np.asarray(self) ** self._np(o)         # noqa: E704
    def __rpow__(self, o): return self._np(o) ** np.asarray(self)        # noqa: E704
    def __neg__(self): return -np.asarray(self)                          # noqa: E704
    def __pos__(self): return +np.asarray(self)                          # noqa: E704
    def __abs__(self): return np.abs(np.asarray(self))                   # noqa: E704
    def __matmul__(self, o): return np.asarray(self) @ self._np(o)       # noqa: E704
    def __rmatmul__(self, o): return self._np(o) @ np.asarray(self)      # noqa: E704
    def __lt__(self, o): return np.asarray(self) < self._np(o)           # noqa: E704
    def __le__(self, o): return np.asarray(self) <= self._np(o)          # noqa: E704
    def __gt__(self, o): return np.asarray(self) > self._np(o)           # noqa: E704
    def __ge__(self, o): return np.asarray(self) >= self._np(o)          # noqa: E704
    # == / != stay identity-based (and the type hashable): leaves are found BY IDENTITY in the tape and in cache signatures;
    # use np.array_equal(a, b) / np.asarray(a) == b for element-wise comparison.


def is_array(x) -> bool:
    return isinstance(x, (np.ndarray, DevArray))


def _children(node) -> List[Tuple[Any, Any]]:
    """(key, child) pairs in flatten order for container nodes; None for leaves."""
    if isinstance(node, Module):
        return [(f, getattr(node, f)) for f in node.__fields__ if hasattr(node, f)]
    if isinstance(node, (list, tuple)):
        return list(enumerate(node))
    if isinstance(node, dict):
        return list(node.items())
    return None


def tree_leaves(tree) -> list:
    """All leaves, depth-first in declaration order (== `jax.tree_util.tree_leaves` on the
    reference's dataclass pytrees; `None` is not a leaf, like in jax)."""
    out = []

    def rec(n):
        ch = _children(n)
        if ch is None:
            if n is not None:
                out.append(n)
            return
        for _, c in ch:
            rec(c)

    rec(tree)
    return out


def _rebuild(n: "Module", rec: Callable, override: Callable = None) -> "Module":
    """New instance of type(n): pytree fields rebuilt IN DECLARATION ORDER (the flatten order that
    `load_torch_weights` relies on -- __init__ may assign them in any order), other attributes shared."""
    new = object.__new__(type(n))
    d = n.__dict__
    for k in n.__fields__:
        if k in d:
            v = d[k]
            object.__setattr__(new, k, override(k, v) if override is not None else rec(v))
    for k, v in d.items():
        if k not in n.__fields__ and k not in ("_dev_cache", "_sig_cache", "_eager_cache"):
            object.__setattr__(new, k, v)
    return new


def tree_map(fn: Callable, tree):
    """Rebuild `tree` with `fn` applied to every leaf (new Module objects, caches dropped)."""

    def rec(n):
        if isinstance(n, Module):
            return _rebuild(n, rec)
        if isinstance(n, list):
            return [rec(c) for c in n]
        if isinstance(n, tuple):
            return tuple(rec(c) for c in n)
        if isinstance(n, dict):
            return {k: rec(c) for k, c in n.items()}
        if n is None:
            return None
        return fn(n)

    return rec(tree)


def tree_replace_leaves(tree, new_leaves: list):
    it = iter(new_leaves)
    out = tree_map(lambda _: next(it), tree)
    return out


def tree_inference(tree, value: bool = True):
    """`eqx.tree_inference`: returns a copy with every field named `inference` set to `value`
    (reference usage: tests/test_models/test_resnet.py:20, README.md:64)."""

    def rec(n):
        if isinstance(n, Module):
            return _rebuild(n, rec, lambda k, v: value if k == "inference" else rec(v))
        if isinstance(n, list):
            return [rec(c) for c in n]
        if isinstance(n, tuple):
            return tuple(rec(c) for c in n)
        if isinstance(n, dict):
            return {k: rec(c) for k, c in n.items()}
        return n      # leaves (arrays, StateIndex, callables, scalars) are shared, like in eqx

    return rec(tree)


_MISSING = object()


def tree_at(where: Callable, tree, replace=_MISSING, replace_fn: Callable = None):
    """Tiny `eqx.tree_at`: `where(tree)` must return one node (or a tuple of nodes); the
    returned tree has them replaced by `replace` (same structure as the targets) or by `replace_fn(node)` (fcn.py:106).
    Implemented by identity search on a deep structural copy."""
    targets = where(tree)
    single = not isinstance(targets, (tuple, list))      # eqx.tree_at takes any sequence of nodes (experimental.py:73-80 passes a list)
    targets = (targets,) if single else tuple(targets)
    if (replace is _MISSING) == (replace_fn is None):
        raise ValueError("tree_at: exactly one of `replace` and `replace_fn` must be given")
    if replace_fn is not None:
        repl = tuple(replace_fn(t) for t in targets)
    else:
        repl = (replace,) if single else tuple(replace)
    ids = {id(t): r for t, r in zip(targets, repl)}

    def rec(n):
        if id(n) in ids:
            return ids[id(n)]
        if isinstance(n, Module):
            return _rebuild(n, rec)
        if isinstance(n, list):
            return [rec(c) for c in n]
        if isinstance(n, tuple):
            return tuple(rec(c) for c in n)
        if isinstance(n, dict):
            return {k: rec(c) for k, c in n.items()}
        return n

    return rec(tree)


def clone(tree):
    return tree_map(lambda x: copy.copy(x) if isinstance(x, StateIndex) else x, tree)
