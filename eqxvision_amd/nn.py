"""`equinox.nn` work-alikes for the layers the reference's hot path instantiates
(SURVEY.md Appendix A): same constructor arguments, same parameter shapes and declaration
order (the `load_torch_weights` contract), single-sample `__call__(x, *, key=None)`.

Computation is delegated to `eqxvision_amd.ops` (C ABI -> HIP kernels).  A call on a raw array
wraps it, runs on the GPU and returns an fp32 torch tensor in the reference's logical layout; a
call on an `Act` (inside `vmap` or inside a model) stays on device in the internal layout.
"""
from __future__ import annotations

import functools
import math
from typing import Any, Callable, Optional, Sequence as Seq, Tuple, Union

import numpy as np

from . import ops
from . import random as jr
from ._act import Act, is_act, wrap
from ._module import Module, StateIndex


# ------------------------------------------------------------------ boundary handling
def _unwrap(out, batched: bool):
    if isinstance(out, Act) and out.node is not None:
        from . import grad as _grad
        if _grad.active():          # inside filter_value_and_grad: the result stays on the tape (grad.GTensor)
            return _grad.GTensor(out)
    if isinstance(out, Act):
        t = ops.to_user(out) if out.kind != "raw" else (out.t if out.batched else out.t[0])
        return t
    if isinstance(out, tuple):
        return tuple(_unwrap(o, batched) for o in out)
    if isinstance(out, list):
        return [_unwrap(o, batched) for o in out]
    return out


def boundary(fn):
    """Let a single-sample `__call__` accept raw arrays (wrap -> run -> unwrap)."""

    @functools.wraps(fn)
    def wrapper(self, x, *args, **kw):
        if is_act(x):
            return fn(self, x, *args, **kw)
        a = wrap(x, batched=False)
        return _unwrap(fn(self, a, *args, **kw), False)

    return wrapper


# ------------------------------------------------------------------ activation functions
def relu(x):
    """jax.nn.relu stand-in (recognised by the fusion peepholes by identity)."""
    if is_act(x):
        return ops.eltwise(x, "relu")
    return _unwrap(ops.eltwise(wrap(x, False), "relu"), False)


def gelu(x):
    """jax.nn.gelu (default approximate=True, tanh form)."""
    if is_act(x):
        return ops.eltwise(x, "gelu")
    return _unwrap(ops.eltwise(wrap(x, False), "gelu"), False)


def _eltwise_fn(name, doc):
    def fn(x):
        if is_act(x):
            return ops.eltwise(x, name)
        return _unwrap(ops.eltwise(wrap(x, False), name), False)
    fn.__name__ = fn.__qualname__ = name
    fn.__doc__ = doc
    return fn


hard_swish = _eltwise_fn("hard_swish", "jax.nn.hard_swish: x * relu6(x + 3) / 6 (mobilenetv3.py:72)")
hard_sigmoid = _eltwise_fn("hard_sigmoid", "jax.nn.hard_sigmoid: relu6(x + 3) / 6 (mobilenetv3.py:57)")
sigmoid = _eltwise_fn("sigmoid", "jax.nn.sigmoid (layers/squeeze.py:40, lraspp.py:102)")
silu = _eltwise_fn("silu", "jax.nn.silu: x * sigmoid(x)")

_ACT_NAMES = {relu: "relu", gelu: "gelu", hard_swish: "hard_swish", hard_sigmoid: "hard_sigmoid", sigmoid: "sigmoid", silu: "silu"}


def act_name(fn) -> Optional[str]:
    """'relu' / 'gelu' if `fn` is one of the fusable activations, else None."""
    try:
        return _ACT_NAMES.get(fn)
    except TypeError:
        return None


def _pair(v) -> Tuple[int, int]:
    if isinstance(v, (tuple, list)):
        if len(v) != 2:
            raise ValueError(f"expected an int or a pair, got {v}")
        return (int(v[0]), int(v[1]))
    return (int(v), int(v))


# ------------------------------------------------------------------ layers
class Identity(Module):
    def __init__(self, *args, **kwargs):
        pass

    def __call__(self, x, *, key=None):
        return x


class Lambda(Module):
    fn: Callable

    def __init__(self, fn: Callable):
        self.fn = fn

    def __call__(self, x, *, key=None):
        return self.fn(x)


class Dropout(Module):
    p: float
    inference: bool

    def __init__(self, p: float = 0.5, inference: bool = False, *, deterministic=None):
        self.p = p
        self.inference = inference if deterministic is None else deterministic

    def __call__(self, x, *, key=None, inference=None):
        inf = self.inference if inference is None else inference
        if inf or self.p == 0:
            return x
        if key is None:                                  # eqx.nn.Dropout: the same RuntimeError
            raise RuntimeError("Dropout requires a key when running in non-deterministic mode.")
        if self.p == 1:
            raise NotImplementedError("Dropout(p=1)")
        if is_act(x):
            return ops.dropout(x, self.p, key)
        return _unwrap(ops.dropout(wrap(x, False), self.p, key), False)


def dropout_live(d) -> bool:
    """A Dropout that drops on this call: training mode and 0 < p."""
    return isinstance(d, Dropout) and not d.inference and d.p > 0


class Conv2d(Module):
    """eqx.nn.Conv2d: weight (out, in/groups, kh, kw), bias (out,1,1)."""

    weight: np.ndarray
    bias: Optional[np.ndarray]
    in_channels: int
    out_channels: int
    kernel_size: Tuple[int, int]
    stride: Tuple[int, int]
    padding: Tuple[int, int]
    dilation: Tuple[int, int]
    groups: int
    use_bias: bool

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, padding=0, dilation=1, groups=1,
                 use_bias=True, *, key=None):
        self.in_channels = int(in_channels)
        self.out_channels = int(out_channels)
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)
        self.dilation = _pair(dilation)
        self.groups = int(groups)
        self.use_bias = bool(use_bias)
        if self.in_channels % self.groups or self.out_channels % self.groups:
            raise ValueError("`in_channels` and `out_channels` must be divisible by `groups`")
        g = jr.generator(key)
        cg = self.in_channels // self.groups
        lim = 1.0 / math.sqrt(cg * self.kernel_size[0] * self.kernel_size[1])
        self.weight = g.uniform(-lim, lim, (self.out_channels, cg) + self.kernel_size).astype(np.float32)
        self.bias = g.uniform(-lim, lim, (self.out_channels, 1, 1)).astype(np.float32) if use_bias else None

    @boundary
    def __call__(self, x, *, key=None):
        if len(x.shape) != 3:
            raise ValueError(f"Input to `Conv2d` needs to have rank 3, but has shape {x.shape}")
        return ops.conv2d(x, self)


class Linear(Module):
    """eqx.nn.Linear: weight (out,in), bias (out,)."""

    weight: np.ndarray
    bias: Optional[np.ndarray]
    in_features: int
    out_features: int
    use_bias: bool

    def __init__(self, in_features, out_features, use_bias=True, *, key=None):
        self.in_features = int(in_features)
        self.out_features = int(out_features)
        self.use_bias = bool(use_bias)
        g = jr.generator(key)
        lim = 1.0 / math.sqrt(self.in_features)
        self.weight = g.uniform(-lim, lim, (self.out_features, self.in_features)).astype(np.float32)
        self.bias = g.uniform(-lim, lim, (self.out_features,)).astype(np.float32) if use_bias else None

    @boundary
    def __call__(self, x, *, key=None):
        if x.kind == "img":          # a (C,H,W) array given to a plain Linear is a rank error in equinox
            raise ValueError(f"Linear expects a vector of {self.in_features} features, got shape {x.shape}")
        return ops.linear(x, self)


class LayerNorm(Module):
    """eqx.nn.LayerNorm(shape, eps=1e-5): weight/bias of `shape`."""

    shape: Tuple[int, ...]
    eps: float
    elementwise_affine: bool
    weight: Optional[np.ndarray]
    bias: Optional[np.ndarray]

    def __init__(self, shape, eps: float = 1e-5, elementwise_affine: bool = True, **kwargs):
        self.shape = (int(shape),) if isinstance(shape, int) else tuple(int(s) for s in shape)
        self.eps = float(eps)
        self.elementwise_affine = bool(elementwise_affine)
        self.weight = np.ones(self.shape, np.float32) if elementwise_affine else None
        self.bias = np.zeros(self.shape, np.float32) if elementwise_affine else None

    @boundary
    def __call__(self, x, *, key=None):
        # under vmap this normalises every row of the feature axis (vit.py:149: jax.vmap(self.norm1)(x))
        return ops.layernorm(x, self)


class BatchNorm(Module):
    """eqx.experimental.BatchNorm(input_size, axis_name, eps=1e-5, momentum=0.99): leaves in the
    reference's order `weight, bias, first_time_index, state_index`; running statistics live in
    `state_index` (set by `load_torch_weights`, reference utils.py:203-218)."""

    weight: Optional[np.ndarray]
    bias: Optional[np.ndarray]
    first_time_index: StateIndex
    state_index: StateIndex
    axis_name: Any
    inference: bool
    input_size: int
    eps: float
    channelwise_affine: bool
    momentum: float

    def __init__(self, input_size, axis_name=None, eps=1e-5, channelwise_affine=True, momentum=0.99,
                 inference=False, **kwargs):
        self.input_size = int(input_size)
        self.axis_name = axis_name
        self.eps = float(eps)
        self.channelwise_affine = bool(channelwise_affine)
        self.momentum = float(momentum)
        self.inference = bool(inference)
        self.weight = np.ones((self.input_size,), np.float32) if channelwise_affine else None
        self.bias = np.zeros((self.input_size,), np.float32) if channelwise_affine else None
        self.first_time_index = StateIndex(True)
        self.state_index = StateIndex(None)

    @boundary
    def __call__(self, x, *, key=None):
        return ops.batchnorm(x, self)


class MaxPool2d(Module):
    kernel_size: Tuple[int, int]
    stride: Tuple[int, int]
    padding: Tuple[int, int]

    def __init__(self, kernel_size, stride=1, padding=0, **kwargs):
        self.kernel_size = _pair(kernel_size)
        self.stride = _pair(stride)
        self.padding = _pair(padding)

    @boundary
    def __call__(self, x, *, key=None):
        return ops.maxpool2d(x, self.kernel_size, self.stride, self.padding)


class AdaptiveAvgPool2d(Module):
    target_shape: Tuple[int, int]

    def __init__(self, target_shape, **kwargs):
        self.target_shape = _pair(target_shape)

    @boundary
    def __call__(self, x, *, key=None):
        return ops.adaptive_avgpool2d(x, self.target_shape)


class Sequential(Module):
    """eqx.nn.Sequential.  `__call__` runs a peephole over the layer list so that
    Conv2d -> BatchNorm -> Lambda(relu) (layers/conv_norm_activation.py:60-85) and
    Conv2d -> Lambda(relu) (alexnet.py:44-55) and Linear -> Lambda(relu) become ONE fused launch."""

    layers: Seq[Module]

    def __init__(self, layers: Seq[Module]):
        self.layers = list(layers)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return Sequential(self.layers[i])
        return self.layers[i]

    def __len__(self):
        return len(self.layers)

    def __iter__(self):
        return iter(self.layers)

    @boundary
    def __call__(self, x, *, key=None):
        return self.call_chained(x, None, key=key)

    def call_chained(self, x, nxt, *, key=None):
        """`__call__`, with `nxt` = the module that will consume the result (or None): the last layer may then fuse
        its tail with `nxt`'s head (see _ResNetBottleneck.call_chained) and attach that result as `.pre`."""
        L = self.layers
        keys = [None] * len(L) if key is None else list(jr.split(key, max(len(L), 1)))
        i = 0
        while i < len(L):
            layer = L[i]
            if type(layer) is Conv2d and is_act(x):
                j = i + 1
                bn = None
                if j < len(L) and isinstance(L[j], BatchNorm) and L[j].inference:
                    bn = L[j]
                    j += 1
                a = None
                if j < len(L) and isinstance(L[j], Lambda) and act_name(L[j].fn):
                    a = act_name(L[j].fn)
                    j += 1
                if x.kind == "img" and a == "relu" and j < len(L) and type(L[j]) is MaxPool2d:
                    # network entry from the raw image followed by a max-pool (alexnet.py:44-46): one launch where the
                    # library has the fused kernel for this configuration (ops.stem_conv_pool falls back to the pair)
                    x = ops.stem_conv_pool(x, layer, bn, a, L[j])
                    i = j + 1
                    continue
                x = ops.conv2d(x, layer, bn, a)
                i = j
                continue
            if type(layer) is Linear and is_act(x) and x.kind != "img":
                j = i + 1
                a = None
                if j < len(L) and isinstance(L[j], Lambda) and act_name(L[j].fn):
                    a = act_name(L[j].fn)
                    j += 1
                x = ops.linear(x, layer, a)
                i = j
                continue
            if hasattr(layer, "call_chained") and is_act(x):
                # consecutive residual blocks (resnet.py:330-333): the block may fuse its tail with the next one's head
                nx = L[i + 1] if i + 1 < len(L) else nxt
                # a nested Sequential (a stage of blocks) takes its share of the key like any other layer (training-mode
                # Dropout / DropPath inside it); a residual block's call_chained has no stochastic layers of its own
                x = layer.call_chained(x, nx, key=keys[i]) if isinstance(layer, Sequential) else layer.call_chained(x, nx)
                i += 1
                continue
            x = layer(x, key=keys[i])
            i += 1
        return x
