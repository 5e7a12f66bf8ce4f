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
from ._act import (Act, _to_device_f32, collect_replay_hooks, compute_dtype, head_fp32, keep_alive, residual_fp32, split_weights, stream_ptr,
                   wrap)
from ._module import DevArray, Module, StateIndex
from .nn import _unwrap


def _is_array(x) -> bool:
    return isinstance(x, (np.ndarray, torch.Tensor))


def vmap(fn: Callable, in_axes=0, out_axes=0, axis_name=None, **_ignored) -> Callable:
    """Map `fn` over axis 0 of its array arguments.  `in_axes` may be an int/None or a tuple with
    one entry per positional argument (None = broadcast, as in tests/test_models/test_vit.py:44).
    `axis_name` only matters for BatchNorm's training-mode `pmean`: the batch axis is physical here, so the batch statistics are
    taken over it directly, and over the data-parallel ranks when the module names an axis (ops.bn_train_update)."""

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
        # `key=keys` (B,2) and other keyword arrays are passed through untouched: in inference they are dead values (modules only
        # check `key is None` like the reference, resnet.py:341-342); training-mode Dropout / DropPath split them per sample
        # (random.split on a [B,2] array) and draw their masks from them.
        out = fn(*call_args, **kwargs)
        return _unwrap(out, True)

    return batched


MAX_INPLACE_VARIANTS = 2   # per signature: graphs that read resident device inputs in place (one per buffer address)


class _Compiled:
    __slots__ = ("static_in", "calls", "keep", "out", "graph", "replays", "refs", "lane_calls", "own_resident", "hooks", "lane_graphs", "done")

    def __init__(self):
        self.own_resident = False      # True: resident device inputs are copied into owned buffers before each replay
        self.static_in = []
        self.lane_calls = None         # lanes > 1: one launch list per sub-batch (c.calls = their concatenation)
        self.calls = []
        self.keep = []
        self.out = None
        self.graph = None
        self.replays = 0
        self.refs = None
        self.hooks = []                # run after every replay (ops register them while recording: _act.on_replay)
        self.lane_graphs = None        # join="stream": one hipGraph per lane, launched on the jitted function's lane streams
        self.done = None               # join="stream": the event each lane records behind its last launch of the latest call


def _is_key_array(x) -> bool:
    """PRNG keys (uint32 arrays): dead values in inference, never shipped to the device."""
    if isinstance(x, np.ndarray):
        return x.dtype == np.uint32
    return isinstance(x, torch.Tensor) and x.dtype in (torch.uint32, torch.int32, torch.int64) and not x.is_cuda


def _is_resident(x) -> bool:
    """A device tensor the kernels can read in place (no staging copy)."""
    return (isinstance(x, torch.Tensor) and x.is_cuda and x.is_contiguous()
            and x.dtype in (torch.float32, torch.bfloat16))


def _module_sig(m: Module):
    """Structural signature of a Module tree: types, field layout, identity of the (immutable, shared) leaf arrays and the
    values of everything else.  Two trees with the same signature run the same launch list on the same weights -- e.g. a
    fresh `tree_inference(net, True)` copy made inside the caller's loop (new Module objects, shared leaves) hits the cache
    instead of retracing.  Cached on the instance: Modules are frozen by convention, like eqx.Module."""
    hit = m.__dict__.get("_sig_cache")
    if hit is not None and all(st.version == v for st, v in hit[1]):
        return hit[0]
    states = []                                # StateIndex slots (BatchNorm running statistics): mutable, shared by identity

    def rec(n):
        if isinstance(n, StateIndex):          # a training-mode step rewrote the statistics -> folded weights / graphs are stale
            states.append((n, n.version))
            return ("state", id(n), n.version)
        if isinstance(n, Module):
            if type(n).__name__ == "BatchNorm" and getattr(n, "inference", True) is False:
                # a training-mode BatchNorm READS AND WRITES its statistics on the device inside the recording: the same launch
                # list serves every step, whatever the version
                return (type(n).__qualname__,) + tuple(
                    (f, ("state-rw", id(getattr(n, f))) if isinstance(getattr(n, f), StateIndex) else rec(getattr(n, f)))
                    for f in n.__fields__ if hasattr(n, f))
            return (type(n).__qualname__,) + tuple((f, rec(getattr(n, f))) for f in n.__fields__ if hasattr(n, f))
        if isinstance(n, (list, tuple)):
            return (type(n).__name__,) + tuple(rec(c) for c in n)
        if isinstance(n, dict):
            return ("dict",) + tuple((k, rec(c)) for k, c in n.items())
        if isinstance(n, (np.ndarray, DevArray)):
            return ("a", id(n), n.shape, str(n.dtype))
        if n is None or isinstance(n, (bool, int, float, str)):
            return n
        return ("o", id(n))                    # StateIndex slots, callables, ...: shared by identity

    sig = ("mod", hash(rec(m)), id(type(m)))
    object.__setattr__(m, "_sig_cache", (sig, states))
    return sig


def _train_layers(x):
    """(BatchNorm modules in training mode, True if a Dropout / DropPath with p > 0 is in training mode) of a Module tree; cached
    on the instance (Modules are frozen; tree_inference builds new ones)."""
    hit = x.__dict__.get("_eager_cache") if isinstance(x, Module) else None
    if hit is not None:
        return hit
    bns, stochastic = [], [False]

    def rec(n):
        if isinstance(n, Module):
            if hasattr(n, "attention_dropout") and (float(n.attention_dropout or 0) > 0 or float(getattr(n, "dropout", 0) or 0) > 0):
                stochastic[0] = True       # Swin's `_func_dropout` draws in EVERY mode (swin.py:17-20)
            if getattr(n, "inference", True) is False:
                if type(n).__name__ == "BatchNorm":
                    bns.append(n)
                elif float(getattr(n, "p", 0.0) or 0.0) > 0.0:
                    stochastic[0] = True
            for f in n.__fields__:
                if hasattr(n, f):
                    rec(getattr(n, f))
        elif isinstance(n, (list, tuple)):
            for c in n:
                rec(c)
        elif isinstance(n, dict):
            for c in n.values():
                rec(c)

    rec(x)
    r = (tuple(bns), stochastic[0])
    if isinstance(x, Module):
        object.__setattr__(x, "_eager_cache", r)
    return r


def _needs_eager(x) -> bool:
    """True when this call of a Module tree cannot be recorded: Dropout / DropPath in training mode (fresh host-derived masks per
    call), a training-mode BatchNorm on its FIRST step (running = batch is a different launch argument than the EMA of later
    steps), or one whose cross-rank sum would go through torch.distributed (not a recorded library call).  Training-mode BatchNorm
    otherwise IS recordable: moments, all-reduce, EMA and fold are stream-ordered library calls (ops.bn_train_update)."""
    bns, stochastic = _train_layers(x)
    if stochastic:
        return True
    if not bns:
        return False
    from . import dist as _dist
    if any(b.first_time_index.value or (b.state_index._dev is None and b.state_index._value is None) for b in bns):
        return True
    return any(b.axis_name is not None for b in bns) and _dist.world_size() > 1 and not _dist._state["native"]


def _has_train_bn(x) -> bool:
    return bool(_train_layers(x)[0])


def _sig(x):
    if _is_array(x):
        if _is_key_array(x):
            return ("key", tuple(x.shape))
        if _is_resident(x):            # read in place; the buffer address selects the variant (see `jitted`)
            return ("dev", tuple(x.shape), str(x.dtype))
        return ("arr", tuple(x.shape), str(x.dtype))
    if isinstance(x, Module):
        return _module_sig(x)
    if callable(x):
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


def _struct_sig(x):
    """`_sig` without leaf identities: types, field layout, leaf shapes / dtypes, static values.  A training loop hands `filter_jit`
    a model with NEW leaves every step (`apply_updates`), so the identity-based signature misses every time; whether a function is
    replayable at all (a training step is not: its loss is a host value) depends on this structure only."""
    if _is_array(x) or isinstance(x, (np.ndarray, DevArray)):
        return ("a", tuple(x.shape), str(x.dtype))
    if isinstance(x, StateIndex):
        return ("state",)
    if isinstance(x, Module):
        return (type(x).__qualname__,) + tuple((f, _struct_sig(getattr(x, f))) for f in x.__fields__ if hasattr(x, f))
    if isinstance(x, (list, tuple)):
        return (type(x).__name__,) + tuple(_struct_sig(v) for v in x)
    if isinstance(x, dict):
        return ("dict",) + tuple((k, _struct_sig(v)) for k, v in sorted(x.items()))
    if callable(x):
        return ("obj", id(x))
    if isinstance(x, (int, float)) and not isinstance(x, bool):
        return ("num", type(x).__name__)       # an optimiser's step counter changes every call; flags (bool / str / None) stay values
    try:
        hash(x)
        return ("val", x)
    except TypeError:
        return ("obj", type(x).__name__)


MAX_CACHE_SIGNATURES = 8   # per jitted function: least-recently-used signatures beyond this are evicted (graph destroyed,
                           # pinned intermediates and argument references dropped)


def _release(c: "_Compiled"):
    if c is None:
        return
    if c.graph is not None:
        try:
            torch.cuda.synchronize()           # nothing may still be replaying the graph / reading its buffers
            _lib.call("mv_graph_destroy", c.graph)
        except Exception:  # noqa: BLE001  (interpreter shutdown)
            pass
        c.graph = None
    if c.lane_graphs:
        try:
            torch.cuda.synchronize()
            for g in c.lane_graphs:
                _lib.call("mv_graph_destroy", g)
        except Exception:  # noqa: BLE001
            pass
        c.lane_graphs, c.done = None, None
    c.calls, c.lane_calls, c.keep, c.static_in, c.refs, c.out = [], None, [], [], None, None


def filter_jit(fn: Callable = None, *, use_graph: bool = True, clone_outputs: bool = True, lanes: int = 1, join: str = "call") -> Callable:
    """`lanes` > 1 (an extension; the reference has no counterpart): the batch is cut into `lanes` contiguous
    sub-batches whose launch lists are captured as PARALLEL branches of the hipGraph (one stream each).  The
    kernels are the same, so are the results; what changes is that the mostly empty last round of CUs of one
    sub-batch's kernel is filled by the other sub-batch's next kernel instead of idling (tile quantisation:
    e.g. 784 / 392 / 196 tiles on 256 CUs for the ResNet-50 3x3 layers at batch 256).

    `join` (with lanes > 1): "call" (default) -- the lanes join inside the call: when it returns, the result is complete in
    stream order on the caller's stream, `jax.jit`'s contract as a torch user reads it.  "stream" (opt-in, round 6) -- each lane has
    its own stream and its own graph and the call returns WITHOUT joining: the next call's first lane starts while this call's last
    lane is still finishing (the tail of a forward -- pooling, classifier -- runs at half occupancy otherwise: resnet50 +2 %, alexnet
    +12 % in tools/lane_offset.py).  That is jax's ASYNCHRONOUS dispatch made explicit: the returned tensors are futures until
    `jitted.ready()` (makes the current stream wait for the latest call; `jitted.block_until_ready()` also synchronises the host).
    Results alternate between TWO sets of buffers, so one stays valid until the second next call; inputs are read in place and must
    not be overwritten before `ready()`.  Requires clone_outputs=False and device-resident inputs."""
    if fn is None:
        return functools.partial(filter_jit, use_graph=use_graph, clone_outputs=clone_outputs, lanes=lanes, join=join)
    if join not in ("call", "stream"):
        raise ValueError(f"filter_jit: join must be 'call' or 'stream', got {join!r}")
    if join == "stream" and (clone_outputs or not use_graph or lanes < 2):
        raise ValueError("filter_jit(join='stream') needs lanes >= 2, use_graph=True and clone_outputs=False (a clone would read the "
                         "result on the caller's stream before the lanes have produced it)")
    pipe = {"streams": None, "last": None}      # join="stream": the lane streams of this function, the entry of the latest call
    import collections
    cache = collections.OrderedDict()
    eager_structs = set()      # argument structures found not replayable (training steps): run eagerly, see `jitted`

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

    def _lane_count(a, kw):
        """lanes if every array argument has the same leading (batch) extent, divisible by `lanes`; else 1."""
        if lanes <= 1 or not use_graph:
            return 1
        if any(_has_train_bn(v) for v in list(a) + list(kw.values()) if isinstance(v, Module)):
            return 1                                 # batch statistics are over the WHOLE batch: no sub-batches
        ext = {int(v.shape[0]) for v in list(a) + list(kw.values()) if _is_array(v) and v.ndim >= 1}
        if len(ext) != 1:
            return 1
        B = ext.pop()
        return lanes if B % lanes == 0 and B // lanes >= 1 else 1

    def _trace_lanes(c, a, kw, nl):
        B = next(int(v.shape[0]) for v in list(a) + list(kw.values()) if _is_array(v))
        step = B // nl
        lane_calls, outs = [], []
        for l in range(nl):
            sl = slice(l * step, (l + 1) * step)
            la = [v[sl] if _is_array(v) else v for v in a]
            lk = {k: (v[sl] if _is_array(v) else v) for k, v in kw.items()}
            calls = []
            old = _lib.set_recording(calls)
            try:
                with keep_alive(c.keep):
                    o = fn(*la, **lk)
            finally:
                _lib.set_recording(old)
            if _lib.not_replayable():
                return                               # host results: not a launch list; the caller's full-batch trace decides
            leaves = list(o) if isinstance(o, (tuple, list)) else [o]
            if not leaves or not all(t is None or (isinstance(t, torch.Tensor) and t.is_cuda and t.shape[0] == step and t.is_contiguous())
                                     for t in leaves) or all(t is None for t in leaves):
                return                               # only batched tensors (or tuples of them / None) are laned; fall back
            lane_calls.append(calls)
            outs.append(o)
        structured = isinstance(outs[0], (tuple, list))
        if structured and any(type(o) is not type(outs[0]) or len(o) != len(outs[0]) or
                              [t is None for t in o] != [t is None for t in outs[0]] for o in outs):
            return
        c.keep.append(outs)

        def gather(lane_leaves, allow_redirect):
            """one full-batch tensor for one output leaf: lane l's rows land in rows l*step .. (l+1)*step"""
            first = lane_leaves[0]
            full = torch.empty((B,) + tuple(first.shape[1:]), dtype=first.dtype, device=first.device)
            for l, o in enumerate(lane_leaves):
                dst = full[l * step:(l + 1) * step]
                # The lane's result is normally the output buffer of exactly ONE launch (the classifier head; the final resize of a
                # segmentation output) and nobody reads it afterwards: point that launch at the lane's rows of the full result
                # instead of copying them there.
                where = [(ci, i) for ci, (_, cargs, _) in enumerate(lane_calls[l]) for i, a in enumerate(cargs)
                         if isinstance(a, int) and a == o.data_ptr()]
                if allow_redirect and len(where) == 1 and o.data_ptr() != 0:
                    ci, ai = where[0]
                    cfn, cargs, cname = lane_calls[l][ci]
                    patched = list(cargs)
                    patched[ai] = dst.data_ptr()
                    lane_calls[l][ci] = (cfn, tuple(patched), cname)
                    rc = cfn(*patched)                       # the trace ran eagerly: produce this call's rows in `full` too
                    if rc != 0:
                        raise _lib.MVError(f"lane output redirect of {cname} failed (rc={rc})")
                    continue
                dt = _lib.F32 if o.dtype == torch.float32 else _lib.BF16     # otherwise: the library's copy kernel
                old = _lib.set_recording(lane_calls[l])
                try:
                    _lib.call("mv_cast", o.data_ptr(), dst.data_ptr(), o.numel(), dt, dt, stream_ptr())
                finally:
                    _lib.set_recording(old)
            return full

        if structured:                                   # e.g. the (aux, out) pair of the segmentation models
            full = type(outs[0])(None if outs[0][i] is None else gather([o[i] for o in outs], True) for i in range(len(outs[0])))
        else:
            full = gather(outs, True)
        c.lane_calls = lane_calls
        c.calls = [x for lc in lane_calls for x in lc]
        c.out = full

    def _replay_list(calls):
        s = stream_ptr()
        for cfn, args, name in calls:
            rc = cfn(*args[:-1], s)
            if rc != 0:
                msg = _lib.load().mv_last_error()
                raise _lib.MVError(f"replay of {name} failed (rc={rc}): {msg.decode() if msg else ''}")

    @functools.wraps(fn)
    def jitted(*args, **kwargs):
        if any(_needs_eager(v) for v in list(args) + list(kwargs.values()) if isinstance(v, (Module, list, tuple, dict))):
            return fn(*args, **kwargs)             # training-mode layers: host logic between launches, nothing to replay
        if eager_structs:                          # this function was traced once with arguments of this STRUCTURE and found not
            sk = (tuple(_struct_sig(a) for a in args), tuple((k, _struct_sig(v)) for k, v in sorted(kwargs.items())))
            if sk in eager_structs:                # replayable (a training step): no signature, no staging copies, no recording, no
                return fn(*args, **kwargs)         # pinned intermediates, no cache entry -- whatever leaves the model carries this step
        key = (compute_dtype(), residual_fp32(), head_fp32(), split_weights(), _lib.load().mv_flags_epoch(), tuple(_sig(a) for a in args),
               tuple((k, _sig(v)) for k, v in sorted(kwargs.items())))
        # A graph bakes buffer addresses in.  Resident device inputs are read in place -- no staging copy -- by a
        # variant per address tuple, at most MAX_INPLACE_VARIANTS of them (a double-buffered loader stays zero-copy);
        # beyond that ONE more variant owns its input buffers and takes a device-to-device copy per call, so a caller
        # that passes a fresh tensor every step neither retraces nor pins a new set of intermediates each time.
        flat_all = list(args) + [v for _, v in sorted(kwargs.items())]
        ptrs = tuple(v.data_ptr() for v in flat_all if _is_array(v) and _is_resident(v))
        grp = cache.get(key)
        if grp is None:
            grp = cache[key] = {"variants": {}, "owned": None}
            while len(cache) > MAX_CACHE_SIGNATURES:         # LRU eviction
                _, old_grp = cache.popitem(last=False)
                for oc in list(old_grp["variants"].values()) + [old_grp["owned"]]:
                    _release(oc)
        else:
            cache.move_to_end(key)
        if join == "stream":                       # two buffer sets per input-address tuple, used in turn
            if len(ptrs) != sum(1 for v in flat_all if _is_array(v) and not _is_key_array(v)):
                raise ValueError("filter_jit(join='stream'): every array argument must be a device-resident tensor (read in place)")
            grp["turn"] = 1 - grp.get("turn", 1)
            ptrs = ptrs + (("turn", grp["turn"]),)
        c = grp["variants"].get(ptrs)
        own = False
        if c is None and ptrs and join != "stream" and len(grp["variants"]) >= MAX_INPLACE_VARIANTS:
            c, own = grp["owned"], True
        elif c is not None:
            own = c.own_resident

        def staged(v):
            return _is_array(v) and not _is_key_array(v) and (own or not _is_resident(v))

        def stage(v):
            return v.clone() if _is_resident(v) else _to_device_f32(v).clone()

        flat_arrays = [a for a in args if staged(a)] + [v for _, v in sorted(kwargs.items()) if staged(v)]
        if c is None:
            c = _Compiled()
            c.own_resident = own
            c.refs = (args, kwargs)            # keep static objects (modules, resident inputs) alive
            new_args, new_kwargs = [], {}
            for a in args:
                if staged(a):                  # owned device staging buffer, refreshed per call
                    t = stage(a)
                    c.static_in.append(t)
                    new_args.append(t)
                else:
                    new_args.append(a)
            for k, v in sorted(kwargs.items()):
                if staged(v):
                    t = stage(v)
                    c.static_in.append(t)
                    new_kwargs[k] = t
                else:
                    new_kwargs[k] = v
            nl = _lane_count(new_args, new_kwargs)
            if nl > 1:
                _trace_lanes(c, new_args, new_kwargs, nl)
            if c.lane_calls is None:
                old = _lib.set_recording(c.calls)
                try:
                    with keep_alive(c.keep), collect_replay_hooks(c.hooks):
                        c.out = fn(*new_args, **new_kwargs)
                finally:
                    _lib.set_recording(old)
            if _lib.not_replayable():
                # the function produced host values while it was traced (a training step: filter_value_and_grad's loss / gradients):
                # what it returned IS this call's result; nothing is cached -- the signature's slot is given back, so a training loop
                # cannot evict compiled inference entries of the same function -- and every later call with arguments of this
                # structure runs eagerly from the top of `jitted` (advisor, round 5: the identity-keyed mark never hit, because
                # apply_updates makes new leaves every step)
                eager_structs.add((tuple(_struct_sig(a) for a in args), tuple((k, _struct_sig(v)) for k, v in sorted(kwargs.items()))))
                if not grp["variants"] and grp["owned"] is None:
                    cache.pop(key, None)
                out = c.out
                _release(c)
                return out
            if own:
                grp["owned"] = c
            else:
                grp["variants"][ptrs] = c
            if join == "stream":
                if c.lane_calls is None:
                    raise ValueError("filter_jit(join='stream'): the function could not be split into lanes (batch not divisible, or "
                                     "outputs that are not batched tensors)")
                pipe["last"] = None                # the trace ran on the caller's stream: complete in stream order
            return _outputs(c)
        # refresh the static input buffers, then replay
        for dst, src in zip(c.static_in, flat_arrays):
            dst.copy_(src if isinstance(src, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(src)),
                      non_blocking=True)
        if join == "stream":
            import ctypes
            cur = torch.cuda.current_stream()
            if pipe["streams"] is None:
                pipe["streams"] = [torch.cuda.Stream() for _ in c.lane_calls]
            if c.lane_graphs is None:              # one graph per lane, captured on that lane's stream
                gs = []
                for st_l, calls in zip(pipe["streams"], c.lane_calls):
                    st_l.wait_stream(cur)
                    with torch.cuda.stream(st_l):
                        _lib.call("mv_graph_begin_capture", stream_ptr())
                        g = ctypes.c_void_p()
                        try:
                            _replay_list(calls)
                        finally:
                            _lib.call("mv_graph_end_capture", stream_ptr(), ctypes.byref(g))
                    gs.append(g)
                c.lane_graphs = gs
                c.done = [torch.cuda.Event() for _ in gs]
            ready = torch.cuda.Event()
            ready.record(cur)                      # everything the caller queued (its writes to the inputs) comes first
            for st_l, g, ev in zip(pipe["streams"], c.lane_graphs, c.done):
                st_l.wait_event(ready)
                with torch.cuda.stream(st_l):
                    _lib.call("mv_graph_launch", g, stream_ptr())
                    ev.record(st_l)
            pipe["last"] = c
            c.replays += 1
            return c.out
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
                    if c.lane_calls is None:
                        _replay(c)
                    else:                              # fork: every extra lane on its own stream, then join
                        fork = torch.cuda.Event()
                        fork.record(side)
                        branches = []
                        for calls in c.lane_calls[1:]:
                            bs = torch.cuda.Stream()
                            bs.wait_event(fork)
                            with torch.cuda.stream(bs):
                                _replay_list(calls)
                                done = torch.cuda.Event()
                                done.record(bs)
                            branches.append((bs, done))
                        _replay_list(c.lane_calls[0])
                        for _, done in branches:
                            side.wait_event(done)
                        c.keep.append(branches)
                finally:
                    _lib.call("mv_graph_end_capture", stream_ptr(), ctypes.byref(g))
            c.graph = g
        if c.graph is not None:
            _lib.call("mv_graph_launch", c.graph, stream_ptr())
        else:
            _replay(c)
        for h in c.hooks:
            h()
        c.replays += 1
        return _outputs(c)

    def ready(stream=None):
        """join="stream": make `stream` (default: the current one) wait for the latest call's lanes."""
        c = pipe["last"]
        if c is not None and c.done:
            st = stream if stream is not None else torch.cuda.current_stream()
            for ev in c.done:
                st.wait_event(ev)

    def block_until_ready():
        ready()
        torch.cuda.current_stream().synchronize()

    jitted.ready = ready
    jitted.block_until_ready = block_until_ready
    jitted._cache = cache
    jitted._eager_structs = eager_structs
    jitted._entries = lambda: [c for g in cache.values()
                               for c in list(g["variants"].values()) + ([g["owned"]] if g["owned"] is not None else [])]
    return jitted
