"""`jax.vmap` / `eqx.filter_jit` work-alikes -- the L4 driver of the reference's hot path:

    @eqx.filter_jit
    def forward(net, images, keys):
        return jax.vmap(net, axis_name="batch")(images, key=keys)      (README.md:37-40)

`vmap` does not trace anything: it marks the leading axis of the array arguments as the batch and
hands `Act`s to the single-sample module code, whose ops are batched natively by the HIP kernels.

`filter_jit` replaces the XLA executable with a recorded launch list: the first call with a given
signature runs the Python body ONCE (eagerly, while every C-ABI call is recorded and every
intermediate buffer is pinned), the second call captures a replay of that list into a hipGraph,
later calls are one `hipGraphLaunch`.  Like `eqx.filter_jit`, arrays are dynamic, everything else
(modules, python values) is static and part of the cache key.
"""
from __future__ import annotations

import functools
from typing import Any, Callable

import numpy as np
import torch

from . import _lib
from ._act import Act, _to_device_f32, compute_dtype, keep_alive, stream_ptr, wrap
from ._module import Module
from .nn import _unwrap


def _is_array(x) -> bool:
    return isinstance(x, (np.ndarray, torch.Tensor))


def vmap(fn: Callable, in_axes=0, out_axes=0, axis_name=None, **_ignored) -> Callable:
    """Map `fn` over axis 0 of its array arguments.  `in_axes` may be an int/None or a tuple with
    one entry per positional argument (None = broadcast, as in tests/test_models/test_vit.py:44).
    `axis_name` only matters for BatchNorm's training-mode `pmean`, which is out of scope."""

    @functools.wraps(fn)
    def batched(*args, **kwargs):
        axes = in_axes if isinstance(in_axes, (tuple, list)) else (in_axes,) * len(args)
        if len(axes) != len(args):
            raise ValueError("vmap in_axes must have one entry per positional argument")
        call_args = []
        for a, ax in zip(args, axes):
            if ax is None or not _is_array(a):
                call_args.append(a)
            elif ax == 0:
                call_args.append(wrap(a, batched=True))
            else:
                raise NotImplementedError("vmap: only in_axes 0/None are supported")
        # `key=keys` (B,2) and other keyword arrays are passed through untouched: in inference they are
        # dead values; modules only check `key is None` like the reference (resnet.py:341-342).
        out = fn(*call_args, **kwargs)
        return _unwrap(out, True)

    return batched


class _Compiled:
    __slots__ = ("static_in", "calls", "keep", "out", "graph", "replays", "refs")

    def __init__(self):
        self.static_in = []
        self.calls = []
        self.keep = []
        self.out = None
        self.graph = None
        self.replays = 0
        self.refs = None


def _is_key_array(x) -> bool:
    """PRNG keys (uint32 arrays): dead values in inference, never shipped to the device."""
    if isinstance(x, np.ndarray):
        return x.dtype == np.uint32
    return isinstance(x, torch.Tensor) and x.dtype in (torch.uint32, torch.int32, torch.int64) and not x.is_cuda


def _is_resident(x) -> bool:
    """A device tensor the kernels can read in place (no staging copy)."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.is_contiguous()
            and x.dtype in (torch.float32, torch.bfloat16))


def _sig(x):
    if _is_array(x):
        if _is_key_array(x):
            return ("key", tuple(x.shape))
        if _is_resident(x):            # read in place: the buffer address is part of the signature
            return ("dev", tuple(x.shape), str(x.dtype), x.data_ptr())
        return ("arr", tuple(x.shape), str(x.dtype))
    if isinstance(x, Module) or callable(x):
        return ("obj", id(x))
    if isinstance(x, (list, tuple)):
        return (type(x).__name__,) + tuple(_sig(v) for v in x)
    if isinstance(x, dict):
        return ("dict",) + tuple((k, _sig(v)) for k, v in sorted(x.items()))
    try:
        hash(x)
        return ("val", x)
    except TypeError:
        return ("obj", id(x))


def filter_jit(fn: Callable = None, *, use_graph: bool = True, clone_outputs: bool = True) -> Callable:
    if fn is None:
        return functools.partial(filter_jit, use_graph=use_graph, clone_outputs=clone_outputs)
    cache = {}

    def _replay(c: _Compiled):
        s = stream_ptr()
        for cfn, args, name in c.calls:
            rc = cfn(*args[:-1], s)
            if rc != 0:
                msg = _lib.load().mv_last_error()
                raise _lib.MVError(f"replay of {name} failed (rc={rc}): {msg.decode() if msg else ''}")

    def _outputs(c: _Compiled):
        def cl(o):
            if isinstance(o, torch.Tensor):
                return o.clone() if clone_outputs else o
            if isinstance(o, (tuple, list)):
                return type(o)(cl(v) for v in o)
            return o
        return cl(c.out)

    @functools.wraps(fn)
    def jitted(*args, **kwargs):
        key = (compute_dtype(), tuple(_sig(a) for a in args), tuple((k, _sig(v)) for k, v in sorted(kwargs.items())))
        c = cache.get(key)

        def staged(v):
            return _is_array(v) and not _is_key_array(v) and not _is_resident(v)

        flat_arrays = [a for a in args if staged(a)] + [v for _, v in sorted(kwargs.items()) if staged(v)]
        if c is None:
            c = _Compiled()
            c.refs = (args, kwargs)            # keep static objects (modules, resident inputs) alive
            new_args, new_kwargs = [], {}
            for a in args:
                if staged(a):                  # host array: owned device staging buffer, refreshed per call
                    t = _to_device_f32(a).clone()
                    c.static_in.append(t)
                    new_args.append(t)
                else:
                    new_args.append(a)
            for k, v in sorted(kwargs.items()):
                if staged(v):
                    t = _to_device_f32(v).clone()
                    c.static_in.append(t)
                    new_kwargs[k] = t
                else:
                    new_kwargs[k] = v
            old = _lib.set_recording(c.calls)
            try:
                with keep_alive(c.keep):
                    c.out = fn(*new_args, **new_kwargs)
            finally:
                _lib.set_recording(old)
            cache[key] = c
            return _outputs(c)
        # refresh the static input buffers, then replay
        for dst, src in zip(c.static_in, flat_arrays):
            dst.copy_(src if isinstance(src, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(src)),
                      non_blocking=True)
        if use_graph and c.graph is None and c.calls:
            # capture on a private stream (the legacy default stream cannot be captured); it first waits
            # for everything already queued on the caller's stream
            import ctypes
            cur = torch.cuda.current_stream()
            side = torch.cuda.Stream()
            side.wait_stream(cur)
            with torch.cuda.stream(side):
                _lib.call("mv_graph_begin_capture", stream_ptr())
                g = ctypes.c_void_p()
                try:
                    _replay(c)
                finally:
                    _lib.call("mv_graph_end_capture", stream_ptr(), ctypes.byref(g))
            c.graph = g
        if c.graph is not None:
            _lib.call("mv_graph_launch", c.graph, stream_ptr())
        else:
            _replay(c)
        c.replays += 1
        return _outputs(c)

    jitted._cache = cache
    return jitted
