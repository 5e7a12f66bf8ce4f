"""Reverse-mode gradients of the forward path + one optimiser: the training step the reference's gradient test takes
(reference tests/test_grads.py:35-47: `eqx.filter_value_and_grad` -> `optax.adam(...).update` -> `eqx.apply_updates`).

    @eqv.filter_value_and_grad
    def compute_loss(model, x, y):
        out = eqv.vmap(model, axis_name="batch")(x, key=keys)
        return eqv.optim.softmax_cross_entropy(out, eqv.optim.one_hot(y, 3)).mean()
    loss, grads = compute_loss(net, images, labels)
    updates, opt_state = optimizer.update(grads, opt_state)
    net = eqv.apply_updates(net, updates)

Design.  While `filter_value_and_grad` runs the function, the hooked entry points of `eqxvision_amd.ops` (HOOKED below) are
routed to the `g_*` functions of this module: the same forward in fp32 from un-fused C-ABI calls, every result carrying a
`Node` (its parents + a closure that turns the result's gradient into the parents' gradients and adds the parameter
gradients to the tape).  The backward pass walks the nodes in reverse creation order.  Every FLOP of both passes is a HIP
kernel behind the C ABI: the contractions of the backward pass that are forward contractions on other operands (Linear dgrad /
wgrad, attention's four products) reuse `mv_linear_fwd` on operands transposed on the device, the rest are the gradient
kernels of csrc/train_bwd.hip.  An un-hooked kernel-enqueueing op inside a differentiated function raises: nothing is silently
left out of the gradient.

BatchNorm.  The reference's training branch normalises with the UPDATED RUNNING statistics, and eqx.experimental.BatchNorm keeps
that state outside the differentiated pytree: the gradient treats the statistics as constants (a per-channel affine).

Scope: the layers of alexnet / vgg / resnet / resnext (basic and bottleneck, un-fused) / mobilenet v2 + v3 / efficientnet / regnet /
vit: Conv2d (any groups), Linear, BatchNorm, LayerNorm, every MV_ACT_* activation, MaxPool2d, AdaptiveAvgPool2d to (1,1) or to the
input size, Dropout, DropPath, SqueezeExcitation (un-fused: pool, two pointwise convolutions, channel scale), attention, cls /
position embeddings, Swin's shifted-window attention (with the relative-position bias table's gradient) and patch merging."""
from __future__ import annotations

import functools
import heapq
import threading
from typing import Callable, Optional

import numpy as np
import torch

from . import _lib
from ._act import Act, device, empty, precision, stream_ptr
from ._module import DevArray, Module, tree_map

_tls = threading.local()
F32 = _lib.F32
ACTS = {None: _lib.ACT_NONE, "none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "gelu": _lib.ACT_GELU_TANH,
        "hard_swish": _lib.ACT_HARD_SWISH, "hard_sigmoid": _lib.ACT_HARD_SIGMOID, "sigmoid": _lib.ACT_SIGMOID, "silu": _lib.ACT_SILU}


# ------------------------------------------------------------------------------------------------------------ the tape
class Tape:
    def __init__(self):
        self.pgrads = {}        # id(leaf) -> device gradient (leaf's shape)
        self.leaves = {}        # id(leaf) -> leaf (kept alive for the duration)
        self.dev = {}           # (id(leaf), tag) -> device copy / re-layout of a parameter
        self.leaf_of = {}       # data_ptr of a device vector handed out by ops.prep_f32 -> leaf
        self.count = 0
        self.inside = 0         # > 0 while a g_* function runs (its own C-ABI calls are allowed)


class Node:
    __slots__ = ("parents", "backward", "order")

    def __init__(self, parents, backward):
        t = tape()
        t.count += 1
        self.order = t.count
        self.parents = list(parents)
        self.backward = backward


def tape() -> Optional[Tape]:
    return getattr(_tls, "tape", None)


def active() -> bool:
    return getattr(_tls, "tape", None) is not None


def _guard(name: str):
    """Called by _lib.call: a kernel launched from an op that has no g_* twin would drop out of the gradient silently."""
    t = tape()
    if t is not None and t.inside == 0 and not name.startswith(("mv_set_flag", "mv_get_flag", "mv_event", "mv_graph", "mv_comm",
                                                                "mv_prng_split", "mv_drop_path_noise")):
        raise NotImplementedError(f"{name} was launched inside filter_value_and_grad by an op without a backward "
                                  "(eqxvision_amd/grad.py lists what is differentiable)")


_lib._grad_guard = _guard


def _op(fn):
    @functools.wraps(fn)
    def wrapper(*a, **kw):
        t = tape()
        t.inside += 1
        try:
            return fn(*a, **kw)
        finally:
            t.inside -= 1
    return wrapper


# ------------------------------------------------------------------------------------------------------------ helpers
def _p(t):
    return None if t is None else t.data_ptr()


def _new(shape):
    return empty(tuple(shape), torch.float32)


def _call(name, *args):
    _lib.call(name, *args)


def _S():
    return stream_ptr()


def _upload(a) -> torch.Tensor:
    if isinstance(a, DevArray):                   # a leaf that already lives on the device (the model has been through apply_updates)
        return a.dev
    return torch.from_numpy(np.ascontiguousarray(np.asarray(a, np.float32))).to(device())


def _leaf_dev(leaf, tag="raw", make=None) -> torch.Tensor:
    """Device fp32 copy of a parameter leaf (tag "raw") or a re-layout of it built by `make(raw)`; cached on the tape."""
    t = tape()
    key = (id(leaf), tag)
    hit = t.dev.get(key)
    if hit is None:
        t.leaves[id(leaf)] = leaf
        if tag == "raw":
            hit = _upload(leaf)
        else:
            hit = make(_leaf_dev(leaf))
        t.dev[key] = hit
    return hit


def _add(a: torch.Tensor, b: torch.Tensor) -> torch.Tensor:
    y = _new(a.shape)
    _call("mv_add_fwd", _p(a), _p(b), _p(y), a.numel(), _lib.ACT_NONE, F32, _S())
    return y


def _acc_param(leaf, g: torch.Tensor):
    t = tape()
    t.leaves[id(leaf)] = leaf
    g = g.reshape(tuple(np.shape(leaf)))
    old = t.pgrads.get(id(leaf))
    t.pgrads[id(leaf)] = g if old is None else _add(old, g)


_BWD_SCRATCH = {}
_BWD_SCRATCH_BYTES = 64 << 20


def _offer_scratch():
    """Hand the next launch on this stream the backward pass's scratch buffer (include/eqxvision_amd.h: mv_set_scratch): the column
    sums and the weight gradients split their reduction over many blocks when they get room for the partial sums.  One buffer per
    stream: the backward pass is eager, and launches on a stream are ordered."""
    key = (_S(), torch.cuda.current_device())
    ws = _BWD_SCRATCH.get(key)
    if ws is None:
        # zero-initialised: the first 4096 bytes are the split-K arrival words of the scratch protocol; the backward entries keep
        # their partial sums behind them, so an offer that a launch leaves unused is still a valid split-K hand-over
        ws = _BWD_SCRATCH[key] = torch.zeros(_BWD_SCRATCH_BYTES, dtype=torch.uint8, device=device())
    _call("mv_set_scratch", _p(ws), _BWD_SCRATCH_BYTES, _S())


def _colsum(a: torch.Tensor, b: Optional[torch.Tensor], C: int) -> torch.Tensor:
    out = _new((C,))
    _offer_scratch()
    _call("mv_colsum_f32", _p(a), _p(b), _p(out), a.numel() // C, C, _S())
    return out


def _transpose(x: torch.Tensor, R: int, C: int, stride: int = 0) -> torch.Tensor:
    y = _new((C, R))
    _call("mv_transpose2d_f32", _p(x), _p(y), R, C, stride, _S())
    return y


def _matmul_nt(x: torch.Tensor, w: torch.Tensor, M: int, N: int, K: int, bias=None) -> torch.Tensor:
    """y[M,N] = x[M,K] . w[N,K]^T (+ bias): the forward Linear entry."""
    y = _new((M, N))
    _call("mv_linear_fwd", _p(x), _p(w), None, _p(bias), None, _p(y), M, N, K, _lib.ACT_NONE, F32, F32, _S())
    return y


def _act_bwd(g: torch.Tensor, ref: torch.Tensor, act) -> torch.Tensor:
    if act in (None, "none"):
        return g
    d = _new(g.shape)
    _call("mv_act_bwd_f32", _p(g), _p(ref), _p(d), g.numel(), ACTS[act], _S())
    return d


def _node(x) -> Optional[Node]:
    return getattr(x, "node", None) if isinstance(x, Act) else None


def _mk(t: torch.Tensor, kind: str, batched: bool, parents, backward) -> Act:
    a = Act(t, kind, batched)
    a.node = Node(parents, backward)
    return a


def _check_act(act):
    if act not in ACTS:
        raise NotImplementedError(f"activation {act!r} has no backward")


# ------------------------------------------------------------------------------------------------------------ layout
@_op
def g_as_map(x: Act) -> Act:
    if x.kind == "map":
        return x
    if x.kind != "img":
        raise ValueError(f"expected an image / feature map, got {x}")
    B, C, H, W = x.t.shape
    y = _new((B, H, W, C))
    _call("mv_nchw_to_nhwc", _p(x.t), _p(y), B, C, H, W, x.dt, F32, _S())
    return Act(y, "map", x.batched)              # the user's image: not differentiated


@_op
def g_as_rows(x: Act, keep_fp32: bool = False) -> Act:
    return x


@_op
def g_cast(x: Act, dtype: str) -> Act:
    return x                                     # everything is fp32 under filter_value_and_grad


@_op
def g_flatten(x: Act) -> Act:
    if x.kind == "vec":
        return x
    if x.kind == "seq":
        shp = tuple(x.t.shape)
        return _mk(x.t.reshape(shp[0], -1), "vec", x.batched, [_node(x)], lambda g: (g.reshape(shp),))
    x = g_as_map.__wrapped__(x)
    B, H, W, C = x.t.shape
    if H * W == 1:
        return _mk(x.t.reshape(B, C), "vec", x.batched, [_node(x)], lambda g: (g.reshape(B, 1, 1, C),))
    y = _new((B, C, H, W))
    _call("mv_nhwc_to_nchw", _p(x.t), _p(y), B, C, H, W, F32, F32, _S())

    def backward(g):
        dx = _new((B, H, W, C))
        _call("mv_nchw_to_nhwc", _p(g), _p(dx), B, C, H, W, F32, F32, _S())
        return (dx,)
    return _mk(y.reshape(B, C * H * W), "vec", x.batched, [_node(x)], backward)


@_op
def g_first_row(x: Act) -> Act:
    B, N, D = x.t.shape
    y = _new((B, D))
    _call("mv_copy_rows", _p(x.t), _p(y), B, 4 * D, 4 * N * D, 4 * D, _S())

    def backward(g):
        dx = torch.zeros((B, N, D), dtype=torch.float32, device=device())
        _call("mv_copy_rows", _p(g), _p(dx), B, 4 * D, 4 * D, 4 * N * D, _S())
        return (dx,)
    return _mk(y, "vec", x.batched, [_node(x)], backward)


# ------------------------------------------------------------------------------------------------------------ element-wise
@_op
def g_eltwise(x: Act, act: str) -> Act:
    _check_act(act)
    if x.kind == "img":
        x = g_as_map.__wrapped__(x)
    y = _new(x.t.shape)
    _call("mv_eltwise_fwd", _p(x.t), _p(y), x.t.numel(), ACTS[act], F32, _S())
    xin = x.t
    return _mk(y, x.kind, x.batched, [_node(x)], lambda g: (_act_bwd(g, xin, act),))


@_op
def g_add(a: Act, b: Act, act=None) -> Act:
    _check_act(act)
    if tuple(a.t.shape) != tuple(b.t.shape):
        raise ValueError(f"add: mismatched operands {a} vs {b}")
    s = _add(a.t, b.t)
    y = s
    if act not in (None, "none"):
        y = _new(s.shape)
        _call("mv_eltwise_fwd", _p(s), _p(y), s.numel(), ACTS[act], F32, _S())

    def backward(g):
        d = _act_bwd(g, s, act)
        return (d, d)
    return _mk(y, a.kind, a.batched, [_node(a), _node(b)], backward)


@_op
def g_dropout(x: Act, p: float, key, per_row: bool = False) -> Act:
    from . import ops
    if x.kind == "img":
        x = g_as_map.__wrapped__(x)
    B, C = x.t.shape[0], x.t.shape[-1]
    if per_row:
        B = x.t.numel() // C
    keys = ops._keys_dev(key, B)
    per = x.t.numel() // B
    chw = 1 if x.kind == "map" else 0
    keep = float(1.0 - p)

    def run(src):
        y = _new(src.shape)
        _call("mv_dropout_fwd", _p(src), _p(keys), _p(y), B, per, C, chw, keep, F32, _S())
        return y
    return _mk(run(x.t), x.kind, x.batched, [_node(x)], lambda g: (run(g),))      # the same mask, the same 1 / keep


@_op
def g_drop_path(x: Act, p: float, mode: str, key) -> Act:
    """DropPath's training branch (drop_path.py:51-61): x * noise with one Bernoulli draw per sample ("global") or per entry of the
    sample's first logical axis ("local"), divided by the keep probability; the gradient is the same scaling."""
    from . import ops
    if float(p) == 0.0:
        return x
    if x.kind == "img":
        x = g_as_map.__wrapped__(x)
    B, C = x.t.shape[0], x.t.shape[-1]
    keep = 1.0 - float(p)
    if x.kind == "seq" and mode != "global":
        raise NotImplementedError("DropPath(mode='local') on a (tokens, features) array is not on any model's path")
    if keep <= 0.0:
        s = torch.zeros((B, C), dtype=torch.float32, device=device())
    else:
        s = _new((B, C))
        _call("mv_drop_path_noise", _p(ops._keys_dev(key, B)), _p(s), B, C, 0 if mode == "global" else 1, float(np.float32(keep)), F32, _S())
    rows = x.t.numel() // (B * C)

    def scale(src):
        y = _new(src.shape)
        _call("mv_channel_scale_nhwc_fwd", _p(src), _p(s), _p(y), B, rows, C, F32, _S())
        return y
    return _mk(scale(x.t), x.kind, x.batched, [_node(x)], lambda g: (scale(g),))


@_op
def g_channel_scale(x: Act, s: Act) -> Act:
    """x * s, s one value per (image, channel) (SqueezeExcitation, layers/squeeze.py:60): dx = g * s, ds = sum_hw g * x."""
    x = g_as_map.__wrapped__(x)
    B, H, W, C = x.t.shape
    st = s.t.reshape(B, -1)
    if st.shape[1] != C or not st.is_contiguous():
        raise ValueError(f"channel_scale: scale {tuple(s.t.shape)} does not fit the map {tuple(x.t.shape)}")
    xin, sshape = x.t, tuple(s.t.shape)

    def scale(src):
        y = _new(src.shape)
        _call("mv_channel_scale_nhwc_fwd", _p(src), _p(st), _p(y), B, H * W, C, F32, _S())
        return y

    def backward(g):
        ds = _new((B, C))
        _call("mv_channel_scale_bwd_f32", _p(g), _p(xin), _p(ds), B, H * W, C, _S())
        return (scale(g), ds.reshape(sshape))
    return _mk(scale(x.t), "map", x.batched, [_node(x), _node(s)], backward)


# ------------------------------------------------------------------------------------------------------------ pooling
@_op
def g_maxpool2d(x: Act, kernel_size, stride, padding) -> Act:
    from .ops import _pair
    x = g_as_map.__wrapped__(x)
    B, H, W, C = x.t.shape
    kh, kw = _pair(kernel_size)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho, Wo = (H + 2 * ph - kh) // sh + 1, (W + 2 * pw - kw) // sw + 1
    y = _new((B, Ho, Wo, C))
    _call("mv_maxpool2d_nhwc_fwd", _p(x.t), _p(y), B, H, W, C, kh, kw, sh, sw, ph, pw, F32, _S())
    xin = x.t

    def backward(g):
        dx = _new((B, H, W, C))
        _call("mv_maxpool2d_bwd_nhwc_f32", _p(xin), _p(g), _p(dx), B, H, W, C, kh, kw, sh, sw, ph, pw, _S())
        return (dx,)
    return _mk(y, "map", x.batched, [_node(x)], backward)


@_op
def g_adaptive_avgpool2d(x: Act, target, out_fp32: bool = False) -> Act:
    from .ops import _pair
    x = g_as_map.__wrapped__(x)
    B, H, W, C = x.t.shape
    oh, ow = _pair(target)
    if oh == H and ow == W:
        return x
    if (oh, ow) != (1, 1):
        raise NotImplementedError(f"AdaptiveAvgPool2d({target}) on a {H}x{W} map has no backward yet (global pooling only)")
    y = _new((B, 1, 1, C))
    _call("mv_adaptive_avgpool2d_nhwc_fwd", _p(x.t), _p(y), B, H, W, C, 1, 1, F32, F32, _S())

    def backward(g):
        dx = _new((B, H, W, C))
        _call("mv_avgpool_global_bwd_nhwc_f32", _p(g), _p(dx), B, H * W, C, _S())
        return (dx,)
    return _mk(y, "map", x.batched, [_node(x)], backward)


# ------------------------------------------------------------------------------------------------------------ BatchNorm
def _bn_vectors(bn, z: torch.Tensor, kind: str, batched: bool):
    """(scale, shift, mean, var) device vectors the normalisation of this call uses: the running statistics after this
    batch's update in training mode (ops.bn_train_update), the stored ones in inference mode."""
    from . import ops
    C = z.shape[-1]
    if not bn.inference:
        sc, sh = ops.bn_train_update(bn, Act(z, kind, batched))
        mean, var = bn.state_index._dev
        # the statistics are updated in place by the next call of this layer: the backward needs the values of THIS call
        mean_, var_ = _new((C,)), _new((C,))
        _call("mv_cast", _p(mean), _p(mean_), C, F32, F32, _S())
        _call("mv_cast", _p(var), _p(var_), C, F32, F32, _S())
        return sc, sh, mean_, var_, dict(bn.state_index._last_update)      # left on the layer's state slot by bn_train_update
    st = bn.state_index.value
    if st is None:
        raise RuntimeError("BatchNorm has no running statistics")
    scale, shift = ops.bn_fold(bn)
    return _upload(scale), _upload(shift), _upload(st[0]), _upload(st[1]), None


def _bn_forward(bn, z: torch.Tensor, kind: str, batched: bool):
    C = z.shape[-1]
    sc, sh, mean, var, train = _bn_vectors(bn, z, kind, batched)
    y = _new(z.shape)
    _call("mv_channel_affine_fwd", _p(z), _p(sc), _p(sh), _p(y), z.numel() // C, C, _lib.ACT_NONE, F32, _S())
    return y, (sc, mean, var, train)


def _bn_backward(bn, z: torch.Tensor, g: torch.Tensor, saved) -> torch.Tensor:
    """dz of y = gamma * (z - mean') * rstd(var') + beta; accumulates dgamma, dbeta.
    Inference mode: mean' / var' are the stored statistics, constants: dz = scale * dy.
    Training mode (eqx.experimental.BatchNorm's training branch): mean' = a * batch_mean + (1 - a) * running_mean (a = 1 on the
    layer's first call, 1 - momentum afterwards), likewise var', and the batch moments are differentiable functions of z -- nothing
    in the reference stops that gradient -- so dz carries the two batch-statistics terms, scaled by a (round-4 advisor finding;
    the oracle differentiates through them too: oracle/torch_grad.py: bn_train).  The column sums run over the whole data-parallel
    batch: summed over ranks when the layer's axis_name spans them."""
    from . import dist as _dist
    sc, mean, var, train = saved
    C = z.shape[-1]
    s1 = s2 = None
    if bn.weight is not None or train is not None:
        s1 = _colsum(g, None, C)
        s2 = _colsum(g, z, C)
    if bn.weight is not None:
        dg = _new((C,))
        _call("mv_bn_dgamma_f32", _p(s2), _p(s1), _p(mean), _p(var), float(bn.eps), _p(dg), C, _S())
        _acc_param(bn.weight, dg)                   # per-rank partial sums, like every other parameter gradient
        _acc_param(bn.bias, s1)
    dz = _new(z.shape)
    zeros = torch.zeros((C,), dtype=torch.float32, device=device())
    if train is None:
        _call("mv_channel_affine_fwd", _p(g), _p(sc), _p(zeros), _p(dz), z.numel() // C, C, _lib.ACT_NONE, F32, _S())
        return dz
    s0 = _colsum(z, None, C)
    if train["reduce"]:
        s1g, s2g = _new((C,)), _new((C,))           # the parameter gradients above keep the local sums
        _call("mv_cast", _p(s1), _p(s1g), C, F32, F32, _S())
        _call("mv_cast", _p(s2), _p(s2g), C, F32, F32, _S())
        for v in (s0, s1g, s2g):
            _dist.all_reduce_sum_(v)
        s1, s2 = s1g, s2g
    A, Bc = _new((C,)), _new((C,))
    _call("mv_bn_train_dz_coef_f32", _p(s1), _p(s2), _p(s0), _p(mean), _p(var), _p(sc), _p(train["cnt"]), float(train["rows"]),
          float(train["a"]), float(bn.eps), _p(A), _p(Bc), C, _S())
    t = _new(z.shape)
    _call("mv_channel_affine_fwd", _p(z), _p(Bc), _p(A), _p(t), z.numel() // C, C, _lib.ACT_NONE, F32, _S())           # A + B z
    _call("mv_channel_affine_res_fwd", _p(g), _p(sc), _p(zeros), _p(t), _p(dz), z.numel() // C, C, _lib.ACT_NONE, F32, _S())
    return dz


@_op
def g_batchnorm(x: Act, bn, act=None) -> Act:
    _check_act(act)
    if x.kind == "seq":
        raise NotImplementedError("BatchNorm on a (tokens, features) array: the reference normalises axis 0; not on the hot path")
    if x.kind == "img":
        x = g_as_map.__wrapped__(x)
    z = x.t
    y1, saved = _bn_forward(bn, z, x.kind, x.batched)
    y = y1
    if act not in (None, "none"):
        y = _new(y1.shape)
        _call("mv_eltwise_fwd", _p(y1), _p(y), y1.numel(), ACTS[act], F32, _S())
    return _mk(y, x.kind, x.batched, [_node(x)], lambda g: (_bn_backward(bn, z, _act_bwd(g, y1, act), saved),))


# ------------------------------------------------------------------------------------------------------------ convolution
def _conv_w(conv) -> torch.Tensor:
    K, C = conv.out_channels, conv.in_channels // conv.groups
    R, S = conv.kernel_size

    def make(raw):
        w = _new((K, R, S, C))
        _call("mv_nchw_to_nhwc", _p(raw), _p(w), K, C, R, S, F32, F32, _S())      # OIHW -> KRSC
        return w
    return _leaf_dev(conv.weight, "krsc", make)


@_op
def g_conv2d(x: Act, conv, bn=None, act=None, residual: Optional[Act] = None) -> Act:
    _check_act(act)
    G = conv.groups
    x = g_as_map.__wrapped__(x)
    B, H, W, C = x.t.shape
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    dh, dw = conv.dilation
    K = conv.out_channels
    if C != conv.in_channels:
        raise ValueError(f"Conv2d expected {conv.in_channels} input channels, got {C}")
    Ho, Wo = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1, (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1
    w = _conv_w(conv)
    bias = _leaf_dev(conv.bias).reshape(-1) if conv.bias is not None else None
    z = _new((B, Ho, Wo, K))
    _call("mv_conv2d_nhwc_fwd", _p(x.t), _p(w), None, _p(bias), None, _p(z), B, H, W, C, K, kh, kw, sh, sw, ph, pw, dh, dw, G,
          _lib.ACT_NONE, F32, F32, _S())
    y1, saved = (z, None) if bn is None else _bn_forward(bn, z, "map", x.batched)
    r = None
    if residual is not None:
        r = g_as_map.__wrapped__(residual)
        if tuple(r.t.shape) != (B, Ho, Wo, K):
            raise ValueError(f"residual shape {tuple(r.t.shape)} != conv output {(B, Ho, Wo, K)}")
    y2 = y1 if r is None else _add(y1, r.t)
    y = y2
    if act not in (None, "none"):
        y = _new(y2.shape)
        _call("mv_eltwise_fwd", _p(y2), _p(y), y2.numel(), ACTS[act], F32, _S())
    xin, need_dx = x.t, _node(x) is not None

    def backward(g):
        g2 = _act_bwd(g, y2, act)
        dz = g2 if bn is None else _bn_backward(bn, z, g2, saved)
        if conv.bias is not None:
            _acc_param(conv.bias, _colsum(dz, None, K))
        dwk = _new((K, kh, kw, C // G))
        _offer_scratch()
        _call("mv_conv2d_wgrad_nhwc_f32", _p(xin), _p(dz), _p(dwk), B, H, W, C, K, kh, kw, sh, sw, ph, pw, dh, dw, G, _S())
        dwo = _new((K, C // G, kh, kw))
        _call("mv_nhwc_to_nchw", _p(dwk), _p(dwo), K, C // G, kh, kw, F32, F32, _S())   # KRSC -> OIHW, the leaf's layout
        _acc_param(conv.weight, dwo)
        dx = None
        if need_dx:
            dx = _new((B, H, W, C))
            _call("mv_conv2d_dgrad_nhwc_f32", _p(dz), _p(w), _p(dx), B, H, W, C, K, kh, kw, sh, sw, ph, pw, dh, dw, G, _S())
        return (dx, g2 if r is not None else None)
    return _mk(y, "map", x.batched, [_node(x), _node(r)], backward)


@_op
def g_stem_conv_pool(x: Act, conv, bn, act, pool) -> Act:
    return g_maxpool2d.__wrapped__(g_conv2d.__wrapped__(x, conv, bn, act), pool.kernel_size, pool.stride, pool.padding)


# ------------------------------------------------------------------------------------------------------------ Linear
@_op
def g_linear(x: Act, lin, act=None, residual: Optional[Act] = None, out_fp32: bool = False) -> Act:
    _check_act(act)
    if x.kind == "img":
        raise ValueError("Linear on a raw image")
    K, N = lin.in_features, lin.out_features
    if x.t.shape[-1] != K:
        raise ValueError(f"Linear expected rows of {K}, got {tuple(x.t.shape)}")
    M = x.t.numel() // K
    W = _leaf_dev(lin.weight)
    bias = _leaf_dev(lin.bias).reshape(-1) if lin.bias is not None else None
    oshape = tuple(x.t.shape[:-1]) + (N,)
    z = _matmul_nt(x.t, W, M, N, K, bias).reshape(oshape)
    if residual is not None and tuple(residual.t.shape) != oshape:
        raise ValueError(f"residual shape {tuple(residual.t.shape)} != Linear output {oshape}")
    y2 = z if residual is None else _add(z, residual.t)
    y = y2
    if act not in (None, "none"):
        y = _new(y2.shape)
        _call("mv_eltwise_fwd", _p(y2), _p(y), y2.numel(), ACTS[act], F32, _S())
    xin, need_dx = x.t, _node(x) is not None

    def backward(g):
        g2 = _act_bwd(g, y2, act)
        if lin.bias is not None:
            _acc_param(lin.bias, _colsum(g2, None, N))
        gT, xT = _transpose(g2, M, N), _transpose(xin, M, K)            # dW[N,K] = g^T . x
        _acc_param(lin.weight, _matmul_nt(gT, xT, N, K, M))
        dx = None
        if need_dx:
            Wt = _leaf_dev(lin.weight, "t", lambda raw: _transpose(raw, N, K))   # [K,N]
            dx = _matmul_nt(g2, Wt, M, K, N).reshape(tuple(xin.shape))         # dx[M,K] = g . W
        return (dx, g2 if residual is not None else None)
    return _mk(y, x.kind, x.batched, [_node(x), _node(residual)], backward)


@_op
def g_linear_head(x: Act, lin) -> Act:
    return g_linear.__wrapped__(x, lin)


# ------------------------------------------------------------------------------------------------------------ LayerNorm
@_op
def g_layernorm(x: Act, ln, out_fp32: bool = False) -> Act:
    if x.kind == "img":
        x = g_as_map.__wrapped__(x)
    C = x.t.shape[-1]
    if int(np.prod(ln.shape)) != C:
        raise ValueError(f"LayerNorm over {ln.shape} applied to rows of {C}")
    gam = _leaf_dev(ln.weight).reshape(-1) if ln.weight is not None else None
    bet = _leaf_dev(ln.bias).reshape(-1) if ln.bias is not None else None
    M = x.t.numel() // C
    y = _new(x.t.shape)
    _call("mv_layernorm_fwd", _p(x.t), _p(gam), _p(bet), _p(y), M, C, 0, float(ln.eps), F32, F32, _S())
    xin = x.t

    def backward(g):
        dx, gx = _new(xin.shape), _new(xin.shape)
        _call("mv_layernorm_bwd_f32", _p(xin), _p(gam), _p(g), _p(dx), _p(gx), M, C, float(ln.eps), _S())
        if ln.weight is not None:
            _acc_param(ln.weight, _colsum(gx, None, C))
            _acc_param(ln.bias, _colsum(g, None, C))
        return (dx,)
    return _mk(y, x.kind, x.batched, [_node(x)], backward)


@_op
def g_ln_linear(x: Act, ln, lin, act=None) -> Act:
    return g_linear.__wrapped__(g_layernorm.__wrapped__(x, ln), lin, act)


@_op
def g_layernorm_first_row(x: Act, ln, out_fp32: bool = False) -> Act:
    return g_layernorm.__wrapped__(g_first_row.__wrapped__(x), ln)


# ------------------------------------------------------------------------------------------------------------ ViT pieces
@_op
def g_prep_f32(mod, name: str, arr):
    """ops.prep_f32 under the tape: the device vector handed to the op is remembered as belonging to leaf `mod.<name>`."""
    if arr is None:
        return None
    leaf = getattr(mod, name)
    t = _leaf_dev(leaf)
    tape().leaf_of[t.data_ptr()] = leaf
    return t.reshape(tuple(np.shape(arr)))


@_op
def g_patch_embed_tokens(x: Act, conv, cls, pos, n_extra: int, out_fp32: bool = False) -> Act:
    """tokens[b, n_extra + p] = conv(x)[b, p] + pos[n_extra + p];  tokens[b, 0] = cls + pos[0]  (vit.py:268-269)."""
    if x.kind != "img":
        raise ValueError("patch_embed expects a raw (C,H,W) image")
    t = tape()
    y = g_conv2d.__wrapped__(x, conv)                                    # [B, Hp, Wp, D] with its own node
    B, Hp, Wp, D = y.t.shape
    P, T = Hp * Wp, n_extra + Hp * Wp
    tok = _new((B, T, D))
    _call("mv_copy_rows", _p(y.t), tok.data_ptr() + 4 * n_extra * D, B, 4 * P * D, 4 * P * D, 4 * T * D, _S())
    if n_extra:
        if n_extra != 1 or cls is None:
            raise NotImplementedError("one cls token only")
        clsr = cls.reshape(1, D)
        for b in range(B):
            _call("mv_copy_rows", _p(clsr), tok.data_ptr() + 4 * b * T * D, 1, 4 * D, 4 * D, 4 * D, _S())
    out = tok
    if pos is not None:
        out = _new((B, T, D))
        for b in range(B):                                                # + pos, broadcast over the batch
            _call("mv_add_fwd", tok.data_ptr() + 4 * b * T * D, _p(pos), out.data_ptr() + 4 * b * T * D, T * D, _lib.ACT_NONE, F32, _S())
    cls_leaf = t.leaf_of.get(cls.data_ptr()) if cls is not None else None
    pos_leaf = t.leaf_of.get(pos.data_ptr()) if pos is not None else None

    def backward(g):                                                      # g [B, T, D]
        if pos_leaf is not None:
            _acc_param(pos_leaf, _colsum(g, None, T * D))                 # sum over the batch
        if cls_leaf is not None:
            rows = _new((B, D))
            _call("mv_copy_rows", _p(g), _p(rows), B, 4 * D, 4 * T * D, 4 * D, _S())
            _acc_param(cls_leaf, _colsum(rows, None, D))
        dy = _new((B, Hp, Wp, D))
        _call("mv_copy_rows", g.data_ptr() + 4 * n_extra * D, _p(dy), B, 4 * P * D, 4 * T * D, 4 * P * D, _S())
        return (dy,)
    return _mk(out, "seq", x.batched, [_node(y)], backward)


@_op
def g_qkv_attention(x: Act, lin, heads: int, scale: float, need_probs: bool, drop=None):
    """qkv Linear + softmax(q k^T scale) v (vit.py:64-73) with the probabilities kept for the backward pass: dP = dO V^T, the softmax
    gradient, dQ = dS K, dK = dS^T Q, dV = P^T dO in two launches (mv_mha_bwd_f32)."""
    if drop is not None:
        raise NotImplementedError("attention dropout has no backward yet")
    B, N, D = x.t.shape
    dh = D // heads
    qkv = g_linear.__wrapped__(x, lin)                                   # [B, N, 3 D], columns [q | k | v][head][dh]
    out = _new((B, N, D))
    probs = _new((B, heads, N, N))
    _call("mv_mha_fwd", _p(qkv.t), _p(out), _p(probs), B, N, heads, dh, float(scale), F32, _S())
    qt = qkv.t

    def backward(g):                                                      # g [B, N, D]
        dqkv = _new((B, N, 3 * D))
        ds = _new((B, heads, N, N))
        _call("mv_mha_bwd_f32", _p(qt), _p(probs), _p(g), _p(ds), _p(dqkv), B, N, heads, dh, float(scale), _S())
        return (dqkv,)
    y = _mk(out, "seq", x.batched, [_node(qkv)], backward)
    return y, (probs if need_probs else None)


# ------------------------------------------------------------------------------------------------------------ Swin pieces
@_op
def g_swin_rel_bias(attn) -> torch.Tensor:
    """table[index] -> [heads][n][n] on the device; the tape remembers which table leaf (and which index) it came from."""
    t = tape()
    table = attn.relative_position_bias_table
    key = (id(table), "relbias")
    hit = t.dev.get(key)
    if hit is None:
        t.leaves[id(table)] = table
        hit = _upload(attn.get_relative_position_bias())
        T = int(np.shape(table)[0])
        idx = (np.asarray(attn.relative_position_index).reshape(-1).astype(np.int64) % T).astype(np.int32)   # negative indices wrap
        t.dev[key] = hit
        t.leaf_of[hit.data_ptr()] = (table, torch.from_numpy(idx).to(device()), T)
    return hit


@_op
def g_swin_window_attention(qkv: Act, bias: torch.Tensor, heads: int, window, shift, drop=None) -> Act:
    if drop is not None:
        raise NotImplementedError("Swin attention dropout has no backward yet")
    B, Hf, Wf, C3 = qkv.t.shape
    C = C3 // 3
    wh, ww, sh, sw = int(window[0]), int(window[1]), int(shift[0]), int(shift[1])
    out = _new((B, Hf, Wf, C))
    _call("mv_swin_window_attn_fwd", _p(qkv.t), _p(bias), _p(out), B, Hf, Wf, C, heads, wh, ww, sh, sw, F32, _S())
    qt = qkv.t
    info = tape().leaf_of.get(bias.data_ptr())

    def backward(g):
        n, nW = wh * ww, (Hf // wh) * (Wf // ww)
        dqkv = _new((B, Hf, Wf, C3))
        gw = _new((B * nW, heads * n * n))
        _call("mv_swin_window_attn_bwd_f32", _p(qt), _p(bias), _p(g), _p(dqkv), _p(gw), B, Hf, Wf, C, heads, wh, ww, sh, sw, _S())
        if info is not None:
            table, idx, T = info
            db = _colsum(gw, None, heads * n * n)                         # [heads][n * n]: summed over images and windows
            dbt = _transpose(db, heads, n * n)                            # [n * n][heads]
            dt = _new((T, heads))
            _call("mv_scatter_rows_sum_f32", _p(dbt), _p(idx), _p(dt), n * n, heads, T, _S())
            _acc_param(table, dt)
        return (dqkv,)
    return _mk(out, "map", qkv.batched, [_node(qkv)], backward)


@_op
def g_patch_merge_gather(x: Act) -> Act:
    x = g_as_map.__wrapped__(x)
    B, H, W, C = x.t.shape
    if H % 2 or W % 2:
        raise NotImplementedError("patch merging of an odd-sized map has no backward yet")
    y = _new((B, H // 2, W // 2, 4 * C))
    _call("mv_patch_merge_gather_nhwc", _p(x.t), _p(y), B, H, W, C, F32, _S())

    def backward(g):
        dx = _new((B, H, W, C))
        _call("mv_patch_merge_gather_bwd_f32", _p(g), _p(dx), B, H, W, C, _S())
        return (dx,)
    return _mk(y, "map", x.batched, [_node(x)], backward)


# ------------------------------------------------------------------------------------------------------------ the transform
HOOKED = ("as_map", "as_rows", "cast", "flatten", "first_row", "eltwise", "add", "dropout", "drop_path", "channel_scale", "maxpool2d",
          "adaptive_avgpool2d", "batchnorm", "conv2d", "stem_conv_pool", "linear", "linear_head", "layernorm", "ln_linear",
          "layernorm_first_row", "prep_f32", "patch_embed_tokens", "qkv_attention", "swin_rel_bias", "swin_window_attention",
          "patch_merge_gather")


def hook(name: str, orig: Callable) -> Callable:
    g = globals()["g_" + name]

    @functools.wraps(orig)
    def routed(*a, **kw):
        if getattr(_tls, "tape", None) is not None:
            return g(*a, **kw)
        return orig(*a, **kw)
    return routed


class GTensor:
    """What `vmap(model)(x)` returns inside `filter_value_and_grad`: the device result with its node.  `.t` is the fp32 device
    tensor (detached values, e.g. for logging)."""

    def __init__(self, act: Act):
        self.act = act

    @property
    def t(self) -> torch.Tensor:
        return self.act.t

    @property
    def shape(self):
        return tuple(self.act.t.shape)


class Scalar:
    """A differentiable device scalar (the loss)."""

    def __init__(self, t: torch.Tensor, node: Optional[Node]):
        self.t, self.node = t, node

    def mean(self):
        return self

    def item(self) -> float:
        return float(self.t.reshape(-1)[0].item())


class _Rows:
    """Per-sample losses; `.mean()` is the scalar the reference differentiates (tests/test_grads.py:41)."""

    def __init__(self, rows: torch.Tensor, mean: torch.Tensor, node: Optional[Node]):
        self.rows, self._mean, self.node = rows, mean, node

    def mean(self) -> Scalar:
        return Scalar(self._mean, self.node)


def one_hot(labels, num_classes: int) -> np.ndarray:
    if isinstance(labels, torch.Tensor):          # under filter_jit the label array arrives staged on the device
        labels = labels.detach().cpu().numpy()
    lab = np.rint(np.asarray(labels)).reshape(-1).astype(np.int64)
    out = np.zeros((lab.shape[0], num_classes), np.float32)
    out[np.arange(lab.shape[0]), lab] = 1.0
    return out


def softmax_cross_entropy(logits, targets) -> _Rows:
    """optax.softmax_cross_entropy(logits, one_hot): per-sample -sum(t * log_softmax(logits)); `.mean()` gives the batch mean.
    `logits`: what `vmap(model)` returned (a GTensor under filter_value_and_grad, else a device / host array)."""
    t = tape()
    act = logits.act if isinstance(logits, GTensor) else None
    lt = act.t if act is not None else (logits if isinstance(logits, torch.Tensor) else _upload(logits)).to(device()).float().contiguous()
    B, K = lt.shape
    tg = _upload(targets).reshape(B, K)
    rows, mean, dl = (torch.empty(n, dtype=torch.float32, device=device()) for n in ((B,), (1,), (B, K)))
    if t is not None:
        t.inside += 1
    try:
        _call("mv_softmax_xent_f32", _p(lt), _p(tg), _p(rows), _p(mean), _p(dl), B, K, _S())
    finally:
        if t is not None:
            t.inside -= 1
    node = None
    if t is not None and act is not None and _node(act) is not None:
        def backward(g):                       # g: the incoming scalar gradient (1 for the loss itself)
            return (dl,)
        node = Node([_node(act)], backward)
    return _Rows(rows, mean, node)


def _backward(root: Node):
    grads = {id(root): (root, None)}           # node id -> (node, gradient); None = the implicit 1 of the scalar loss
    heap = [(-root.order, id(root))]
    seen = {id(root)}
    t = tape()
    t.inside += 1                              # the backward closures and the gradient accumulation launch kernels of their own
    try:
        while heap:
            _, nid = heapq.heappop(heap)
            node, g = grads.pop(nid)
            pg = node.backward(g)
            for parent, gp in zip(node.parents, pg):
                if parent is None or gp is None:
                    continue
                if id(parent) in grads:
                    grads[id(parent)] = (parent, _add(grads[id(parent)][1], gp))
                else:
                    grads[id(parent)] = (parent, gp)
                if id(parent) not in seen:
                    seen.add(id(parent))
                    heapq.heappush(heap, (-parent.order, id(parent)))
    finally:
        t.inside -= 1


def filter_value_and_grad(fn: Callable) -> Callable:
    """`eqx.filter_value_and_grad`: fn(model, *args) -> scalar loss; returns (loss, grads) with `grads` = the model tree with
    every array leaf replaced by its gradient (zeros where the loss does not depend on it) and every other leaf kept."""

    @functools.wraps(fn)
    def wrapped(model, *args, **kwargs):
        if active():
            raise RuntimeError("filter_value_and_grad does not nest")
        # under filter_jit (the reference wraps its make_step in eqx.filter_jit): the loss is a host float and the gradients are read
        # after a synchronize -- a replay of the launches would hand back the first call's values.  The trace in progress is marked
        # so that filter_jit runs this function eagerly on every call (transforms.jitted).
        _lib.mark_not_replayable("filter_value_and_grad returns host values")
        if not torch.cuda.is_available():
            raise _lib.MVError("eqxvision_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
        t = _tls.tape = Tape()
        try:
            with precision("fp32"):
                loss = fn(model, *args, **kwargs)
                if isinstance(loss, _Rows):
                    loss = loss.mean()
                if not isinstance(loss, Scalar) or loss.node is None:
                    raise TypeError("filter_value_and_grad: the function must return a scalar that depends on the model "
                                    "(e.g. eqv.optim.softmax_cross_entropy(vmap(model)(x), targets).mean())")
                _backward(loss.node)
                torch.cuda.synchronize()

                def leaf_grad(leaf):
                    if not _float_leaf(leaf):
                        return leaf
                    g = t.pgrads.get(id(leaf))
                    if g is None:
                        g = torch.zeros(tuple(leaf.shape), dtype=torch.float32, device=device())
                    return DevArray(g.contiguous())
                grads = tree_map(leaf_grad, model)
                return loss.item(), grads
        finally:
            _tls.tape = None
    return wrapped


# ------------------------------------------------------------------------------------------------------------ optimiser
def _float_leaf(x) -> bool:
    return isinstance(x, (np.ndarray, DevArray)) and x.dtype.kind == "f"


class _Adam:
    """optax.adam(learning_rate, b1, b2, eps): init(params) -> state; update(grads, state) -> (updates, state)."""

    def __init__(self, learning_rate: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8):
        self.lr, self.b1, self.b2, self.eps = float(learning_rate), float(b1), float(b2), float(eps)

    def init(self, params):
        from ._module import tree_leaves
        n = [int(np.prod(l.shape)) for l in tree_leaves(params) if _float_leaf(l)]
        z = lambda k: torch.zeros(k, dtype=torch.float32, device=device())
        return {"count": 0, "mu": [z(k) for k in n], "nu": [z(k) for k in n]}

    def update(self, grads, state, params=None):
        count = state["count"] + 1
        bc1, bc2 = 1.0 - self.b1 ** count, 1.0 - self.b2 ** count
        it = iter(range(len(state["mu"])))

        def step(leaf):
            if not _float_leaf(leaf):
                return leaf
            i = next(it)
            g = _upload(leaf)
            upd = torch.empty(g.numel(), dtype=torch.float32, device=device())
            _lib.call("mv_adam_step_f32", _p(g), _p(state["mu"][i]), _p(state["nu"][i]), _p(upd), g.numel(), self.lr, self.b1, self.b2,
                      self.eps, bc1, bc2, stream_ptr())
            return DevArray(upd.reshape(tuple(leaf.shape)))
        updates = tree_map(step, grads)
        return updates, {"count": count, "mu": state["mu"], "nu": state["nu"]}


def adam(learning_rate: float, b1: float = 0.9, b2: float = 0.999, eps: float = 1e-8) -> _Adam:
    return _Adam(learning_rate, b1, b2, eps)


def apply_updates(model, updates):
    """`eqx.apply_updates`: a new model whose float array leaves are leaf + update, everything else shared.  The sum is taken on the
    device and STAYS there: the new leaves are `DevArray`s (the analogue of the reference's jax.Array leaves), so a training loop
    moves its parameters across PCIe once -- the first step's upload -- and its gradients, moments and updates never
    (`np.asarray(leaf)` / `state_dict` fetch a host copy on demand).  Round 4 downloaded every leaf here and uploaded it again in
    the next step: 5 crossings of the parameter set per step (alexnet B = 8: 118 ms)."""
    from ._module import tree_leaves
    ups = [u for u in tree_leaves(updates) if _float_leaf(u)]
    it = iter(ups)

    def step(leaf):
        if not _float_leaf(leaf):
            return leaf
        u = next(it)
        if leaf.dtype != np.float32:               # the device state is fp32 only: other float widths keep their dtype, summed on the host
            return (np.asarray(leaf) + np.asarray(u).reshape(leaf.shape).astype(leaf.dtype)).astype(leaf.dtype)
        ud = _upload(u)
        p = _upload(leaf)
        out = torch.empty_like(p)
        _lib.call("mv_add_fwd", _p(p), _p(ud.reshape(p.shape)), _p(out), p.numel(), _lib.ACT_NONE, F32, stream_ptr())
        return DevArray(out.reshape(tuple(leaf.shape)))
    return tree_map(step, model)
