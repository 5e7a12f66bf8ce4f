"""Host-side operator layer: Python wrappers that marshal `Act`s into the C ABI
(`include/eqxvision_amd.h`).  One function per fused op of the hot path; weight preparation
(BN folding, KRSC re-layout, bf16 conversion) happens once per module and is cached on it.

Everything here enqueues HIP kernels on torch's current stream; nothing computes on the host.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch

from . import _lib
from ._act import (Act, is_act, DT, TORCH_DT, compute_dtype, device, empty, head_fp32, on_replay, residual_fp32, split_weights,
                   stream_ptr)

ACT = {None: _lib.ACT_NONE, "none": _lib.ACT_NONE, "relu": _lib.ACT_RELU, "gelu": _lib.ACT_GELU_TANH,
       "hard_swish": _lib.ACT_HARD_SWISH, "hard_sigmoid": _lib.ACT_HARD_SIGMOID, "sigmoid": _lib.ACT_SIGMOID, "silu": _lib.ACT_SILU}
UNFUSED_ACTS = ("hard_swish", "hard_sigmoid", "sigmoid", "silu")    # element-wise entries + depthwise conv only (header: MV_ACT_*)


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else t.data_ptr()


def _dev(a: np.ndarray, dtype: torch.dtype) -> torch.Tensor:
    a = np.ascontiguousarray(a)
    if dtype == torch.bfloat16 and a.dtype == np.float32 and a.size < (1 << 20):
        # small per-call constants (DropPath noise, folded scales): round to bf16 with integer numpy -- a torch CPU cast wakes the
        # whole intra-op thread pool, 16 ms per call on a 100+-core host against microseconds here
        u = a.view(np.uint32)
        r = ((u >> np.uint32(16)) & np.uint32(1)) + np.uint32(0x7FFF)
        b = np.where(np.isnan(a), np.uint32(0x7FC0), (u + r) >> np.uint32(16)).astype(np.uint16)
        return torch.from_numpy(b.view(np.int16)).to(device()).view(torch.bfloat16)
    t = torch.from_numpy(a)
    if t.dtype != dtype:
        t = t.to(dtype)            # host-side RNE conversion of constants (weights)
    return t.to(device())


def _pair(v):
    return tuple(v) if isinstance(v, (tuple, list)) else (v, v)


# ------------------------------------------------------------------ weight preparation (cached)
def bn_fold(bn) -> Tuple[np.ndarray, np.ndarray]:
    """BatchNorm inference as per-channel scale/shift:  (x-mean)/sqrt(var+eps)*w+b."""
    st = bn.state_index.value
    if st is None:
        raise RuntimeError(
            "BatchNorm has no running statistics (the reference's unloaded eqx.experimental.BatchNorm "
            "cannot run inference either); load weights or set them via load_torch_weights")
    mean, var = st
    inv = 1.0 / np.sqrt(np.asarray(var, np.float32) + np.float32(bn.eps))
    g = np.asarray(bn.weight, np.float32) if bn.weight is not None else np.ones_like(inv)
    b = np.asarray(bn.bias, np.float32) if bn.bias is not None else np.zeros_like(inv)
    scale = (g * inv).astype(np.float32)
    shift = (b - np.asarray(mean, np.float32) * scale).astype(np.float32)
    return scale, shift


def _bn_training(bn) -> bool:
    return bn is not None and not bn.inference


def _bn_id(bn):
    """Cache-key part for weights prepared WITH a BatchNorm folded in: identity + the version of its running statistics (the
    StateIndex is shared by every copy of the module, e.g. tree_inference's; a training-mode update bumps it, so an inference
    call afterwards refolds)."""
    return None if bn is None else (id(bn), bn.state_index.version)


def bn_train_update(bn, y: Act):
    """The statistics half of eqx.experimental.BatchNorm's TRAINING branch (resnet.py:132-136 with the model
    not in inference mode): per-channel batch mean, then the mean of squared deviations from it -- two passes, each summed over
    the ranks of `axis_name`'s data-parallel group (RCCL all-reduce of C floats on the launch stream) --, then the running
    statistics: first call: running = batch; later: running = (1 - momentum) * batch + momentum * running.  Everything stays on
    the device and in stream order (no host round trip inside a step); the host `StateIndex.value` is fetched when asked for.
    Returns the (scale, shift) device vectors of the UPDATED running statistics: what the caller normalises with (reference
    semantics)."""
    from . import dist as _dist
    t = y.t
    C = t.shape[-1]
    if C != bn.input_size:
        raise ValueError(f"BatchNorm({bn.input_size}) applied to {C} channels")
    rows = t.numel() // C
    if not _lib.load().mv_channel_moments_supported(rows, C, y.dt):
        raise NotImplementedError(f"training-mode BatchNorm over {C} channels (multiples of 8 up to 2048 only)")
    ws = empty((int(_lib.load().mv_channel_moments_ws(C)),), torch.float32)
    s, q, mean = (empty((C,), torch.float32) for _ in range(3))
    reduce_ranks = bn.axis_name is not None and _dist.world_size() > 1
    cnt = None
    st = stream_ptr()
    if reduce_ranks:                       # rows of the GLOBAL batch (shards may be ragged): summed on the device
        mine, cnt = empty((1,), torch.float32), empty((1,), torch.float32)
        mine.fill_(float(rows))            # (a constant of the recording; the copy + in-place sum below are what a replay repeats)
        _lib.call("mv_cast", _ptr(mine), _ptr(cnt), 1, _lib.F32, _lib.F32, st)
        _dist.all_reduce_sum_(cnt)
    sidx = bn.state_index
    first = bool(bn.first_time_index.value) or (sidx._dev is None and sidx._value is None)
    if sidx._dev is None:                  # the running statistics move to the device once and stay there
        host = sidx._value
        run = (torch.zeros(C, dtype=torch.float32, device=device()), torch.ones(C, dtype=torch.float32, device=device())) \
            if host is None else tuple(_dev(np.asarray(a, np.float32), torch.float32) for a in host)
    else:
        run = sidx._dev
    w = prep_f32(bn, "weight", bn.weight) if bn.weight is not None else None
    b = prep_f32(bn, "bias", bn.bias) if bn.bias is not None else None
    scale, shift = empty((C,), torch.float32), empty((C,), torch.float32)
    if first or _lib.get_flag("bn_two_pass"):
        # the reference's literal two passes (mean, then squared deviations from it): nothing to centre a single pass on yet
        _lib.call("mv_channel_moments_fwd", _ptr(t), None, _ptr(s), _ptr(ws), rows, C, 0, y.dt, st)
        if reduce_ranks:
            _dist.all_reduce_sum_(s)
        _lib.call("mv_bn_mean_fwd", _ptr(s), _ptr(cnt), float(rows), _ptr(mean), C, st)
        _lib.call("mv_channel_moments_fwd", _ptr(t), _ptr(mean), _ptr(q), _ptr(ws), rows, C, 1, y.dt, st)
        if reduce_ranks:
            _dist.all_reduce_sum_(q)
        _lib.call("mv_bn_ema_fold_fwd", _ptr(q), _ptr(mean), _ptr(cnt), float(rows), _ptr(run[0]), _ptr(run[1]), _ptr(w), _ptr(b),
                  _ptr(scale), _ptr(shift), float(bn.momentum), float(bn.eps), 1 if first else 0, C, st)
    else:
        # steady state: both moments about the running mean in ONE pass, ONE all-reduce of 2 x C floats per layer
        ws2, sums = empty((2 * ws.numel(),), torch.float32), empty((2 * C,), torch.float32)
        _lib.call("mv_channel_moments2_fwd", _ptr(t), _ptr(run[0]), _ptr(sums), _ptr(ws2), rows, C, y.dt, st)
        if reduce_ranks:
            _dist.all_reduce_sum_(sums)
        _lib.call("mv_bn_ema_fold1_fwd", _ptr(sums), _ptr(cnt), float(rows), _ptr(run[0]), _ptr(run[1]), _ptr(w), _ptr(b),
                  _ptr(scale), _ptr(shift), float(bn.momentum), float(bn.eps), C, st)
    if first:
        bn.first_time_index.value = False
    # what the backward of THIS call needs (grad._bn_backward): the weight of the batch statistics in the statistics the layer
    # normalised with (running' = a * batch + (1 - a) * running), the row count of the data-parallel batch and whether the column
    # sums have to be summed over ranks
    sidx._last_update = dict(a=1.0 if first else 1.0 - float(bn.momentum), rows=float(rows), cnt=cnt, reduce=reduce_ranks)
    sidx.device_updated(run)               # bumps the version: every fold prepared with the old statistics is stale (_bn_id)
    on_replay(lambda s=sidx, r=run: s.device_updated(r))     # ... and again after every replay of a recorded step
    return scale, shift


def _bn_affine(y: Act, scale: torch.Tensor, shift: torch.Tensor, act=None, residual: Optional[Act] = None) -> Act:
    C = y.t.shape[-1]
    out = empty(tuple(y.t.shape), y.t.dtype)
    if residual is not None and C % 8 == 0 and residual.t.dtype == y.t.dtype and tuple(residual.t.shape) == tuple(y.t.shape):
        _lib.call("mv_channel_affine_res_fwd", _ptr(y.t), _ptr(scale), _ptr(shift), _ptr(residual.t), _ptr(out), y.t.numel() // C, C,
                  ACT[act], y.dt, stream_ptr())
        return Act(out, y.kind, y.batched)
    _lib.call("mv_channel_affine_fwd", _ptr(y.t), _ptr(scale), _ptr(shift), _ptr(out), y.t.numel() // C, C,
              ACT[act if residual is None else None], y.dt, stream_ptr())
    z = Act(out, y.kind, y.batched)
    return z if residual is None else add(z, residual, act)


def prep_conv(conv, bn, layout: str, dtype: str):
    key = ("conv", layout, dtype, _bn_id(bn))
    cache = conv._cache()
    hit = cache.get(key)
    if hit is not None:
        return hit
    w = np.asarray(conv.weight, np.float32)                      # (O, I/g, kh, kw)
    if layout == "krsc":
        w = np.ascontiguousarray(w.transpose(0, 2, 3, 1))
    bias = None if conv.bias is None else np.asarray(conv.bias, np.float32).reshape(-1)
    scale = shift = None
    if bn is not None:
        scale, shift = bn_fold(bn)
        if bias is not None:
            shift = shift + bias * scale
    elif bias is not None:
        shift = bias
    out = (_dev(w, TORCH_DT[dtype]),
           None if scale is None else _dev(scale, torch.float32),
           None if shift is None else _dev(shift, torch.float32))
    cache[key] = out
    return out


def prep_conv_grouped64(conv, bn):
    """Grouped filters (O, I/g, kh, kw) expanded to the per-tile input windows: [O][kh][kw][win] (the layout
    mv_conv2d_nhwc_grouped64_fwd documents; block-diagonal 64 x 64 tiles when the group width divides 64), bf16; BN folded to
    fp32 scale / shift as in prep_conv."""
    key = ("conv_g64", _bn_id(bn))
    cache = conv._cache()
    hit = cache.get(key)
    if hit is not None:
        return hit
    w = np.asarray(conv.weight, np.float32)                      # (O, Cg, kh, kw)
    O_, cg, kh, kw = w.shape
    win = int(_lib.load().mv_conv2d_grouped64_window(O_, conv.groups))
    w64 = np.zeros((O_, kh, kw, win), np.float32)
    for k in range(O_):
        off = (k // cg) * cg - ((k // 64 * 64) // cg) * cg       # my group's first channel inside my tile's window
        w64[k, :, :, off:off + cg] = w[k].transpose(1, 2, 0)
    bias = None if conv.bias is None else np.asarray(conv.bias, np.float32).reshape(-1)
    scale = shift = None
    if bn is not None:
        scale, shift = bn_fold(bn)
        if bias is not None:
            shift = shift + bias * scale
    elif bias is not None:
        shift = bias
    hit = (_dev(w64, torch.bfloat16), None if scale is None else _dev(scale, torch.float32),
           None if shift is None else _dev(shift, torch.float32))
    cache[key] = hit
    return hit


def prep_linear(lin, dtype: str):
    key = ("lin", dtype)
    cache = lin._cache()
    hit = cache.get(key)
    if hit is not None:
        return hit
    w = _dev(np.asarray(lin.weight, np.float32), TORCH_DT[dtype])
    b = None if lin.bias is None else _dev(np.asarray(lin.bias, np.float32).reshape(-1), torch.float32)
    cache[key] = (w, b)
    return w, b


def _bf16_split(w: np.ndarray):
    """fp32 -> (hi, lo) bf16 tensors with hi + lo ~ w to ~16 mantissa bits (host, round-to-nearest-even)."""
    t = torch.from_numpy(np.ascontiguousarray(w.astype(np.float32)))
    hi = t.to(torch.bfloat16)
    lo = (t - hi.to(torch.float32)).to(torch.bfloat16)
    return hi, lo


def prep_linear_split(lin):
    """[N][2K] = [hi row | lo row] bf16 + fp32 bias (cached)."""
    cache = lin._cache()
    hit = cache.get("lin_split")
    if hit is None:
        hi, lo = _bf16_split(np.asarray(lin.weight, np.float32))
        w = torch.cat([hi, lo], dim=1).contiguous().to(device())
        b = None if lin.bias is None else _dev(np.asarray(lin.bias, np.float32).reshape(-1), torch.float32)
        hit = (w, b)
        cache["lin_split"] = hit
    return hit


def prep_conv_split(conv):
    """OIHW weights as (hi, lo) bf16 + fp32 bias (cached); no BatchNorm on these call sites."""
    cache = conv._cache()
    hit = cache.get("conv_split")
    if hit is None:
        hi, lo = _bf16_split(np.asarray(conv.weight, np.float32))
        b = None if conv.bias is None else _dev(np.asarray(conv.bias, np.float32).reshape(-1), torch.float32)
        hit = (hi.to(device()), lo.to(device()), b)
        cache["conv_split"] = hit
    return hit


def prep_f32(mod, name: str, arr) -> Optional[torch.Tensor]:
    if arr is None:
        return None
    cache = mod._cache()
    key = ("f32", name)
    hit = cache.get(key)
    if hit is None:
        from ._module import DevArray
        hit = arr.dev.reshape(-1) if isinstance(arr, DevArray) else _dev(np.asarray(arr, np.float32), torch.float32)
        cache[key] = hit
    return hit


# ------------------------------------------------------------------ layout plumbing
def as_map(x: Act) -> Act:
    """img (NCHW, user dtype) -> map (NHWC, compute dtype)."""
    if x.kind == "map":
        return x
    if x.kind != "img":
        raise ValueError(f"expected an image / feature map, got {x}")
    B, C, H, W = x.t.shape
    dt = compute_dtype()
    y = empty((B, H, W, C), TORCH_DT[dt])
    _lib.call("mv_nchw_to_nhwc", _ptr(x.t), _ptr(y), B, C, H, W, x.dt, DT[dt], stream_ptr())
    return Act(y, "map", x.batched)


def as_rows(x: Act, keep_fp32: bool = False) -> Act:
    """seq / vec in the compute dtype (fp32 rows pass through when `keep_fp32`: residual streams)."""
    dt = compute_dtype()
    if x.t.dtype == TORCH_DT[dt] or (keep_fp32 and x.t.dtype == torch.float32):
        return x
    y = empty(tuple(x.t.shape), TORCH_DT[dt])
    _lib.call("mv_cast", _ptr(x.t), _ptr(y), x.t.numel(), x.dt, DT[dt], stream_ptr())
    return Act(y, x.kind, x.batched)


def to_user(x: Act) -> torch.Tensor:
    """Act -> fp32 torch tensor in the reference's logical layout (batch axis kept iff batched)."""
    if x.ln is not None:            # a residual stream kept as two bf16 planes never leaves as its high plane alone
        x = stream_f32(x)
    if x.kind == "map":
        B, H, W, C = x.t.shape
        y = empty((B, C, H, W), torch.float32)
        _lib.call("mv_nhwc_to_nchw", _ptr(x.t), _ptr(y), B, C, H, W, x.dt, _lib.F32, stream_ptr())
    elif x.t.dtype == torch.float32:
        y = x.t
    else:
        y = empty(tuple(x.t.shape), torch.float32)
        _lib.call("mv_cast", _ptr(x.t), _ptr(y), x.t.numel(), x.dt, _lib.F32, stream_ptr())
    return y if x.batched else y[0]


def flatten(x: Act) -> Act:
    """jnp.ravel of the logical (C,H,W) sample -> vec, CHW order (alexnet.py:83, resnet.py:355)."""
    if x.kind == "vec":
        return x
    if x.kind == "seq":
        B = x.B
        return Act(x.t.reshape(B, -1), "vec", x.batched)
    x = as_map(x)
    B, H, W, C = x.t.shape
    if H * W == 1:
        return Act(x.t.reshape(B, C), "vec", x.batched)
    y = empty((B, C, H, W), x.t.dtype)
    _lib.call("mv_nhwc_to_nchw", _ptr(x.t), _ptr(y), B, C, H, W, x.dt, x.dt, stream_ptr())
    return Act(y.reshape(B, C * H * W), "vec", x.batched)


# ------------------------------------------------------------------ contractions
STEM_MAX_CIN = 4


def conv2d(x: Act, conv, bn=None, act=None, residual: Optional[Act] = None) -> Act:
    """Conv2d [+ BatchNorm(inference)] [+ residual] [+ relu/gelu], one launch.  A BatchNorm in TRAINING mode cannot be folded:
    convolution, batch statistics (+ cross-rank sum), running-statistics update, normalisation, (+ residual, activation)."""
    if _bn_training(bn):
        y = conv2d(x, conv, None, None, None)
        sc, sh = bn_train_update(bn, y)
        return _bn_affine(y, sc, sh, act, None if residual is None else as_map(residual))
    dt = compute_dtype()
    if act in UNFUSED_ACTS and (dt != "bf16" or (x.kind != "img" and conv.out_channels % 8) or
                                (conv.groups > 1 and not conv.groups == conv.in_channels == conv.out_channels)):
        # hard_swish & co. are fused by the NHWC bf16 convolution (dense: the streaming / one tile kernel have them; depthwise: in
        # the kernel) and by the image-entry kernels; the grouped / padded-width paths and fp32 mode take an element-wise pass
        return eltwise(conv2d(x, conv, bn, None, residual), act)
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    dh, dw = conv.dilation
    K, Cin = conv.out_channels, conv.in_channels
    if x.kind == "img" and Cin <= STEM_MAX_CIN and conv.groups == 1 and dh == 1 and dw == 1 and residual is None:
        B, C, H, W = x.t.shape
        if C != Cin:
            raise ValueError(f"Conv2d expected {Cin} input channels, got {C}")
        w, scale, shift = prep_conv(conv, bn, "oihw", dt)
        Ho = (H + 2 * ph - (kh - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - (kw - 1) - 1) // sw + 1
        y = empty((B, Ho, Wo, K), TORCH_DT[dt])
        _lib.call("mv_conv2d_nchw_fwd", _ptr(x.t), _ptr(w), _ptr(scale), _ptr(shift), _ptr(y),
                  B, C, H, W, K, kh, kw, sh, sw, ph, pw, ACT[act], x.dt, DT[dt], 0, 0, None, stream_ptr())
        return Act(y, "map", x.batched)
    x = as_map(x)
    B, H, W, C = x.t.shape
    if C != Cin:
        raise ValueError(f"Conv2d expected {Cin} input channels, got {C}")
    if x.t.dtype != TORCH_DT[dt]:
        raise ValueError(f"activation dtype {x.t.dtype} does not match compute dtype {dt}")
    Ho = (H + 2 * ph - dh * (kh - 1) - 1) // sh + 1
    Wo = (W + 2 * pw - dw * (kw - 1) - 1) // sw + 1

    def res_ptr():
        if residual is None:
            return None
        r = as_map(residual)
        if tuple(r.t.shape) != (B, Ho, Wo, K):
            raise ValueError(f"residual shape {tuple(r.t.shape)} != conv output {(B, Ho, Wo, K)}")
        return _ptr(r.t)

    if conv.groups > 1 and dt == "bf16" and residual is None and \
            _lib.load().mv_dwconv2d_supported(C, K, conv.groups, kh, kw, _lib.BF16, _lib.BF16):
        cache = conv._cache()
        hit = cache.get(("dw", _bn_id(bn)))
        if hit is None:
            _, scale, shift = prep_conv(conv, bn, "oihw", dt)
            wr = np.ascontiguousarray(np.asarray(conv.weight, np.float32)[:, 0].transpose(1, 2, 0))     # (C,1,R,S) -> [R][S][C]
            hit = (_dev(wr, torch.bfloat16), scale, shift)
            cache[("dw", _bn_id(bn))] = hit
        y = empty((B, Ho, Wo, K), torch.bfloat16)
        _lib.call("mv_dwconv2d_nhwc_fwd", _ptr(x.t), _ptr(hit[0]), _ptr(hit[1]), _ptr(hit[2]), _ptr(y), B, H, W, C, kh, kw, sh, sw,
                  ph, pw, dh, dw, ACT[act], _lib.BF16, _lib.BF16, stream_ptr())
        return Act(y, "map", x.batched)
    if conv.groups > 1 and dt == "bf16" and _lib.load().mv_conv2d_grouped64_supported(C, K, kh, kw, conv.groups, _lib.BF16, _lib.BF16):
        w64, scale, shift = prep_conv_grouped64(conv, bn)
        y = empty((B, Ho, Wo, K), torch.bfloat16)
        _lib.call("mv_conv2d_nhwc_grouped64_fwd", _ptr(x.t), _ptr(w64), _ptr(scale), _ptr(shift), res_ptr(), _ptr(y),
                  B, H, W, C, K, kh, kw, sh, sw, ph, pw, dh, dw, conv.groups, ACT[act], _lib.BF16, _lib.BF16, stream_ptr())
        return Act(y, "map", x.batched)
    w, scale, shift = prep_conv(conv, bn, "krsc", dt)
    if K % 8 and dt == "bf16" and conv.groups == 1 and residual is None and C % 8 == 0:
        # an output width the MFMA kernels cannot store in 16-byte pieces (21-class segmentation heads, fcn.py:33): run the
        # convolution on zero-padded filters and compact the rows afterwards instead of dropping to the VALU kernel
        return _conv2d_padded_k(x, conv, bn, act, (w, scale, shift), (B, H, W, C, Ho, Wo))
    y = empty((B, Ho, Wo, K), TORCH_DT[dt])
    if conv.groups == 1 and dt == "bf16":
        _splitk_scratch(B * Ho * Wo, K, kh * kw * C)
    _lib.call("mv_conv2d_nhwc_fwd", _ptr(x.t), _ptr(w), _ptr(scale), _ptr(shift), res_ptr(), _ptr(y),
              B, H, W, C, K, kh, kw, sh, sw, ph, pw, dh, dw, conv.groups, ACT[act], DT[dt], DT[dt], stream_ptr())
    return Act(y, "map", x.batched)


def _conv2d_padded_k(x: Act, conv, bn, act, prep, dims) -> Act:
    w, scale, shift = prep
    B, H, W, C, Ho, Wo = dims
    K = conv.out_channels
    Kp = (K + 7) // 8 * 8
    cache = conv._cache()
    key = ("kpad", _bn_id(bn))
    hit = cache.get(key)
    if hit is None:
        wp = torch.zeros((Kp,) + tuple(w.shape[1:]), dtype=w.dtype, device=w.device)
        wp[:K] = w
        pad1 = lambda v, fill: None if v is None else torch.cat([v, torch.full((Kp - K,), fill, dtype=v.dtype, device=v.device)])
        hit = (wp, pad1(scale, 1.0), pad1(shift, 0.0))
        cache[key] = hit
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    dh, dw = conv.dilation
    yp = empty((B, Ho, Wo, Kp), torch.bfloat16)
    _lib.call("mv_conv2d_nhwc_fwd", _ptr(x.t), _ptr(hit[0]), _ptr(hit[1]), _ptr(hit[2]), None, _ptr(yp),
              B, H, W, C, Kp, kh, kw, sh, sw, ph, pw, dh, dw, 1, ACT[act], _lib.BF16, _lib.BF16, stream_ptr())
    y = empty((B, Ho, Wo, K), torch.bfloat16)
    _lib.call("mv_copy_rows", _ptr(yp), _ptr(y), B * Ho * Wo, K * 2, Kp * 2, K * 2, stream_ptr())
    return Act(y, "map", x.batched)


def _pointwise(conv) -> bool:
    return (tuple(conv.kernel_size) == (1, 1) and tuple(conv.stride) == (1, 1) and tuple(conv.padding) == (0, 0)
            and tuple(conv.dilation) == (1, 1) and conv.groups == 1)


def conv1x1_chain(x: Act, conv3, bn3, residual: Act, conv1n, bn1n, sub: int = 0) -> Optional[Act]:
    """relu(bn3(conv3(x)) + residual), and in the same launch relu(bn1n(conv1n(.))) of that result: the tail of one
    ResNet bottleneck and the head of the next (resnet.py:144-162).  Returns the first result with the second attached
    as `.pre = (conv1n, Act)`, or None when the library has no fused path for the shapes (caller falls back to conv2d).
    `sub` = 2: the caller guarantees that the only other consumer of the first result is a stride-2 pointwise convolution
    (the next stage's downsample branch); where the library can, only the pixels that convolution reads are written
    (`.sub = 2`, a quarter of the map); otherwise the full map is written as usual."""
    dt = compute_dtype()
    if dt != "bf16" or not (_pointwise(conv3) and _pointwise(conv1n)) or conv3.out_channels != conv1n.in_channels:
        return None
    if _bn_training(bn3) or _bn_training(bn1n):
        return None
    x, residual = as_map(x), as_map(residual)
    B, H, W, C = x.t.shape
    K, N2 = conv3.out_channels, conv1n.out_channels
    M = B * H * W
    if C != conv3.in_channels or tuple(residual.t.shape) != (B, H, W, K) or x.t.dtype != torch.bfloat16 \
            or residual.t.dtype != torch.bfloat16:
        return None
    if not _lib.load().mv_conv1x1_chain_supported(M, C, K, N2, DT[dt]):
        return None
    lib = _lib.load()
    for sb in ((2, 0) if sub == 2 else (0,)):      # the accumulator-layout kernel (csrc/chain_rc.hip: chain_res) where it has the shape
        if lib.mv_conv1x1_chain_res_supported(B, H, W, C, K, N2, sb, DT[dt]):
            wf, shf, _ = chain_res_fragments(conv3, bn3, conv1n, bn1n)
            y = empty((B, H // 2, W // 2, K) if sb else (B, H, W, K), torch.bfloat16)
            t1 = empty((B, H, W, N2), torch.bfloat16)
            _lib.call("mv_conv1x1_chain_res_fwd", _ptr(x.t), _ptr(residual.t), _ptr(wf), _ptr(shf), _ptr(y), _ptr(t1), B, H, W, C, K, N2,
                      sb, DT[dt], stream_ptr())
            out = Act(y, "map", x.batched)
            out.sub = 2 if sb else None
            out.pre = (conv1n, Act(t1, "map", x.batched))
            return out
    w3, s3, h3 = prep_conv(conv3, bn3, "krsc", dt)
    w1, s1, h1 = prep_conv(conv1n, bn1n, "krsc", dt)
    t1 = empty((B, H, W, N2), torch.bfloat16)
    if sub == 2 and _lib.load().mv_conv1x1_chain_sub_supported(B, H, W, C, K, N2, DT[dt]):
        y = empty((B, H // 2, W // 2, K), torch.bfloat16)
        _lib.call("mv_conv1x1_chain_sub_fwd", _ptr(x.t), _ptr(w3), _ptr(s3), _ptr(h3), _ptr(residual.t), _ptr(y), _ptr(w1), _ptr(s1),
                  _ptr(h1), _ptr(t1), B, H, W, C, K, N2, DT[dt], stream_ptr())
        out = Act(y, "map", x.batched)
        out.sub = 2
    else:
        y = empty((B, H, W, K), torch.bfloat16)
        _lib.call("mv_conv1x1_chain_fwd", _ptr(x.t), _ptr(w3), _ptr(s3), _ptr(h3), _ptr(residual.t), _ptr(y), _ptr(w1), _ptr(s1),
                  _ptr(h1), _ptr(t1), M, C, K, N2, DT[dt], stream_ptr())
        out = Act(y, "map", x.batched)
    out.pre = (conv1n, Act(t1, "map", x.batched))
    return out


def _rc_shift_rows(*vecs) -> np.ndarray:
    """The shift rows of mv_conv1x1_chain_rc*_fwd: per 32 values one row of 64 uint32 words, word r < 32 = bf16 hi(v) | bf16 lo(v) << 16
    (two bf16 terms: the shift goes through the matrix pipe as one more k-step), words 32 .. 63 zero."""
    rows = []
    for v in vecs:
        v = np.asarray(v, np.float32).reshape(-1, 32)
        hi = torch.from_numpy(v).to(torch.bfloat16)
        lo = (torch.from_numpy(v) - hi.to(torch.float32)).to(torch.bfloat16)
        w = hi.view(torch.int16).numpy().astype(np.uint16).astype(np.uint32) | (lo.view(torch.int16).numpy().astype(np.uint16).astype(np.uint32) << 16)
        out = np.zeros((v.shape[0], 64), np.uint32)
        out[:, :32] = w
        rows.append(out)
    return np.concatenate(rows, 0)


def _rc_conv1n_frags(frags, base, w1n, c):
    for fh in range(2):
        for s_ in range(2):
            cols = 32 * c + 16 * s_ + 4 * fh + np.array([0, 1, 2, 3, 8, 9, 10, 11])
            for a2 in range(w1n.shape[0] // 32):
                frags[c, base + 2 * s_ + a2, fh] = w1n[32 * a2:32 * a2 + 32][:, cols]


def chain_rc_fragments(conv3_0, bn3_0, ds_conv, ds_bn, conv3_1, bn3_1, conv1n, bn1n):
    """Operands of mv_conv1x1_chain_rc_fwd (header): per 32-channel chunk c of the block output, 16 MFMA A fragments
    [lane = 32 fh + r][8] -- 8 of [scale3 W3_0 | scale_d W_d][32 c + r, 16 kk + 8 fh ..], 4 of scale1 W3_1[32 c + r, ...], 4 of the
    scaled next conv1 (k-step s, row tile a2) with the reduction index in accumulator order -- and the 18 shift rows (cached on conv3_1)."""
    cache = conv3_1._cache()
    key = ("chain_rc", _bn_id(bn3_1), id(conv3_0), _bn_id(bn3_0), id(ds_conv), _bn_id(ds_bn), id(conv1n), _bn_id(bn1n))
    hit = cache.get(key)
    if hit is None:
        w30, h30 = _scaled_rows(conv3_0, bn3_0)
        wd, hd = _scaled_rows(ds_conv, ds_bn)
        wcat = np.concatenate([w30, wd], axis=1)                                  # [256][128], as ops._dual_weights
        w31, h1 = _scaled_rows(conv3_1, bn3_1)                                    # [256][64]
        w1n, hn = _scaled_rows(conv1n, bn1n)                                      # [64][256]
        K = wcat.shape[0]
        frags = np.empty((K // 32, 16, 2, 32, 8), np.float32)                      # (chunk, fragment, fh, r, e)
        e8 = np.arange(8)
        for c in range(K // 32):
            rows = slice(32 * c, 32 * c + 32)
            for fh in range(2):
                for kk in range(8):
                    frags[c, kk, fh] = wcat[rows][:, 16 * kk + 8 * fh + e8]
                for kk in range(4):
                    frags[c, 8 + kk, fh] = w31[rows][:, 16 * kk + 8 * fh + e8]
            _rc_conv1n_frags(frags, 12, w1n, c)
        sh = _rc_shift_rows(h30 + hd, h1, hn)
        hit = (_dev(frags.reshape(-1), torch.bfloat16), torch.from_numpy(sh.view(np.int32)).to(device()), (conv3_0, ds_conv, conv1n))
        cache[key] = hit
    return hit


def chain_res_fragments(conv3, bn3, conv1n, bn1n):
    """mv_conv1x1_chain_res_fwd's operands: per 32-channel chunk 4 fragments of scale3 W3 and 2 N2 / 32 of the scaled next conv1
    (k-step s, row tile a2; accumulator order), then the shift rows shift3 | shiftN (cached on conv3)."""
    cache = conv3._cache()
    key = ("chain_res", _bn_id(bn3), id(conv1n), _bn_id(bn1n))
    hit = cache.get(key)
    if hit is None:
        w3, h3 = _scaled_rows(conv3, bn3)
        w1n, hn = _scaled_rows(conv1n, bn1n)
        K, T2 = w3.shape[0], w1n.shape[0] // 32
        frags = np.empty((K // 32, 4 + 2 * T2, 2, 32, 8), np.float32)
        e8 = np.arange(8)
        for c in range(K // 32):
            rows = slice(32 * c, 32 * c + 32)
            for fh in range(2):
                for kk in range(4):
                    frags[c, kk, fh] = w3[rows][:, 16 * kk + 8 * fh + e8]
                for s_ in range(2):
                    cols = 32 * c + 16 * s_ + 4 * fh + np.array([0, 1, 2, 3, 8, 9, 10, 11])
                    for a2 in range(T2):
                        frags[c, 4 + T2 * s_ + a2, fh] = w1n[32 * a2:32 * a2 + 32][:, cols]
        sh = _rc_shift_rows(h3, hn)
        hit = (_dev(frags.reshape(-1), torch.bfloat16), torch.from_numpy(sh.view(np.int32)).to(device()), conv1n)
        cache[key] = hit
    return hit


def chain_rc0_fragments(conv3_0, bn3_0, ds_conv, ds_bn, conv1n, bn1n):
    """mv_conv1x1_chain_rc0_fwd's operands: 12 fragments per 32-channel chunk (the 8 of [scale3 W3_0 | scale_d W_d], the 4 of the scaled
    next conv1 in accumulator order) and the 10 shift rows (cached on conv3_0)."""
    cache = conv3_0._cache()
    key = ("chain_rc0", _bn_id(bn3_0), id(ds_conv), _bn_id(ds_bn), id(conv1n), _bn_id(bn1n))
    hit = cache.get(key)
    if hit is None:
        w30, h30 = _scaled_rows(conv3_0, bn3_0)
        wd, hd = _scaled_rows(ds_conv, ds_bn)
        wcat = np.concatenate([w30, wd], axis=1)
        w1n, hn = _scaled_rows(conv1n, bn1n)
        K = wcat.shape[0]
        frags = np.empty((K // 32, 12, 2, 32, 8), np.float32)
        e8 = np.arange(8)
        for c in range(K // 32):
            rows = slice(32 * c, 32 * c + 32)
            for fh in range(2):
                for kk in range(8):
                    frags[c, kk, fh] = wcat[rows][:, 16 * kk + 8 * fh + e8]
            _rc_conv1n_frags(frags, 8, w1n, c)
        sh = _rc_shift_rows(h30 + hd, hn)
        hit = (_dev(frags.reshape(-1), torch.bfloat16), torch.from_numpy(sh.view(np.int32)).to(device()), (ds_conv, conv1n))
        cache[key] = hit
    return hit


def conv1x1_chain_rc(t2: Act, t2_prev: Act, x0: Act, conv3_0, bn3_0, ds_conv, ds_bn, conv3_1, bn3_1, conv1n, bn1n) -> Optional[Act]:
    """The boundary between the second and the third bottleneck of a stage whose FIRST block output was not written
    (ops.conv1x1_dual_chain(..., store_y=False)): y0 = relu(bn3_0(conv3_0(t2_prev)) + ds_bn(ds_conv(x0))) is recomputed from its
    two 64-channel sources, then y1 = relu(bn3_1(conv3_1(t2)) + y0) and relu(bn1n(conv1n(y1))) as ops.conv1x1_chain
    (resnet.py:144-162, 295-303).  Returns y1 with the next conv1's result attached as `.pre`, or None if unsupported."""
    dt = compute_dtype()
    if dt != "bf16" or not all(_pointwise(c) for c in (conv3_0, ds_conv, conv3_1, conv1n)):
        return None
    if any(_bn_training(b) for b in (bn3_0, ds_bn, bn3_1, bn1n)):
        return None
    t2, t2_prev, x0 = as_map(t2), as_map(t2_prev), as_map(x0)
    B, H, W, C = t2.t.shape
    K, N2 = conv3_1.out_channels, conv1n.out_channels
    M = B * H * W
    if tuple(t2_prev.t.shape) != (B, H, W, C) or tuple(x0.t.shape) != (B, H, W, C) or conv3_0.in_channels != C \
            or ds_conv.in_channels != C or conv3_1.in_channels != C or conv3_0.out_channels != K or ds_conv.out_channels != K \
            or conv1n.in_channels != K or any(a.t.dtype != torch.bfloat16 for a in (t2, t2_prev, x0)):
        return None
    if not _lib.load().mv_conv1x1_chain_rc_supported(M, C, K, N2, DT[dt]):
        return None
    wf, tab, _ = chain_rc_fragments(conv3_0, bn3_0, ds_conv, ds_bn, conv3_1, bn3_1, conv1n, bn1n)
    y = empty((B, H, W, K), torch.bfloat16)
    t1 = empty((B, H, W, N2), torch.bfloat16)
    _lib.call("mv_conv1x1_chain_rc_fwd", _ptr(t2.t), _ptr(t2_prev.t), _ptr(x0.t), _ptr(wf), _ptr(tab), _ptr(y), _ptr(t1), M, C, K, N2,
              DT[dt], stream_ptr())
    out = Act(y, "map", t2.batched)
    out.pre = (conv1n, Act(t1, "map", t2.batched))
    return out


def _fold(conv, bn):
    """fp32 (scale, shift) of a convolution's bias + BatchNorm(inference) epilogue (ones / zeros where absent)."""
    K = conv.out_channels
    bias = None if conv.bias is None else np.asarray(conv.bias, np.float32).reshape(-1)
    if bn is not None:
        scale, shift = bn_fold(bn)
        if bias is not None:
            shift = shift + bias * scale
        return scale, shift
    return np.ones(K, np.float32), (np.zeros(K, np.float32) if bias is None else bias)


def prep_bneck_tail(conv2, bn2, conv3, bn3):
    """Weights of a bottleneck's 3x3 and expansion convolutions in the FRAGMENT ORDER of mv_bottleneck_tail_fwd (header): a wave
    owns 32 output channels and fetches each k16-step of them as one contiguous 1 KB piece (cached on conv2)."""
    key = ("bneck_tail", _bn_id(bn2), id(conv3), _bn_id(bn3))
    cache = conv2._cache()
    hit = cache.get(key)
    if hit is None:
        w2 = np.ascontiguousarray(np.asarray(conv2.weight, np.float32).transpose(0, 2, 3, 1))     # [K][R][S][C]
        K, R, S, C = w2.shape
        w2f = w2.reshape(K // 32, 32, R * S, C // 16, 2, 8).transpose(0, 2, 3, 4, 1, 5)            # (wave, tap, j, h, m, e)
        w3 = np.asarray(conv3.weight, np.float32).reshape(conv3.out_channels, C)
        w3f = w3.reshape(conv3.out_channels // 256, 8, 32, C // 16, 2, 8).transpose(0, 1, 3, 4, 2, 5)   # (chunk, wave, j, h, m, e)
        s2, h2 = _fold(conv2, bn2)
        s3, h3 = _fold(conv3, bn3)
        hit = (_dev(w2f, torch.bfloat16), _dev(s2, torch.float32), _dev(h2, torch.float32),
               _dev(w3f, torch.bfloat16), _dev(s3, torch.float32), _dev(h3, torch.float32), conv3)   # conv3: keeps the id() in the key alive
        cache[key] = hit
    return hit


def bottleneck_tail(t1: Act, conv2, bn2, conv3, bn3, identity: Act) -> Optional[Act]:
    """relu(bn3(conv3(relu(bn2(conv2(t1))))) + identity) in ONE launch with the `width`-channel intermediate resident in LDS,
    one workgroup per image (resnet.py:144-162, identity blocks).  None when the library has no such path for the shapes."""
    dt = compute_dtype()
    if dt != "bf16" or _bn_training(bn2) or _bn_training(bn3) or not _pointwise(conv3):
        return None
    if tuple(conv2.kernel_size) != (3, 3) or tuple(conv2.stride) != (1, 1) or tuple(conv2.padding) != (1, 1) \
            or tuple(conv2.dilation) != (1, 1) or conv2.groups != 1 or conv2.in_channels != conv2.out_channels \
            or conv3.in_channels != conv2.out_channels:
        return None
    t1, identity = as_map(t1), as_map(identity)
    B, H, W, C = t1.t.shape
    K = conv3.out_channels
    if C != conv2.in_channels or tuple(identity.t.shape) != (B, H, W, K) or t1.t.dtype != torch.bfloat16 \
            or identity.t.dtype != torch.bfloat16:
        return None
    if not _lib.load().mv_bottleneck_tail_supported(H, W, C, K, DT[dt]):
        return None
    w2f, s2, h2, w3f, s3, h3, _ = prep_bneck_tail(conv2, bn2, conv3, bn3)
    y = empty((B, H, W, K), torch.bfloat16)
    _lib.call("mv_bottleneck_tail_fwd", _ptr(t1.t), _ptr(w2f), _ptr(s2), _ptr(h2), _ptr(w3f), _ptr(s3), _ptr(h3),
              _ptr(identity.t), _ptr(y), B, H, W, C, K, DT[dt], stream_ptr())
    return Act(y, "map", t1.batched)


def _scaled_rows(conv, bn) -> Tuple[np.ndarray, np.ndarray]:
    """A pointwise conv + BatchNorm(inference) as (scale[k] * W[k, :], shift[k]) in fp32."""
    w = np.asarray(conv.weight, np.float32).reshape(conv.out_channels, -1)
    bias = None if conv.bias is None else np.asarray(conv.bias, np.float32).reshape(-1)
    if bn is not None:
        scale, shift = bn_fold(bn)
        if bias is not None:
            shift = shift + bias * scale
        return w * scale[:, None], shift
    return w, (np.zeros(conv.out_channels, np.float32) if bias is None else bias)


def _dual_weights(conv3, bn3, ds_conv, ds_bn):
    """[scale3 * W3 | scale_d * W_d] as bf16 rows and shift3 + shift_d (cached on conv3)."""
    cache = conv3._cache()
    key = ("dual", _bn_id(bn3), id(ds_conv), _bn_id(ds_bn))
    hit = cache.get(key)
    if hit is None:
        w3, h3 = _scaled_rows(conv3, bn3)
        wd, hd = _scaled_rows(ds_conv, ds_bn)
        hit = (_dev(np.concatenate([w3, wd], axis=1), torch.bfloat16), _dev((h3 + hd).astype(np.float32), torch.float32))
        cache[key] = hit
    return hit


def conv1x1_dual(x: Act, conv3, bn3, xin: Act, ds_conv, ds_bn, act="relu") -> Optional[Act]:
    """act(bn3(conv3(x)) + ds_bn(ds_conv(xin))): a bottleneck's last conv and its (possibly strided) pointwise downsample
    branch as one GEMM over the concatenated reduction (resnet.py:144-162, 295-303).  None when unsupported."""
    dt = compute_dtype()
    if dt != "bf16" or not _pointwise(conv3):
        return None
    sd = tuple(ds_conv.stride)
    if tuple(ds_conv.kernel_size) != (1, 1) or tuple(ds_conv.padding) != (0, 0) or tuple(ds_conv.dilation) != (1, 1) \
            or ds_conv.groups != 1 or sd[0] != sd[1] or conv3.out_channels != ds_conv.out_channels:
        return None
    if _bn_training(bn3) or _bn_training(ds_bn):
        return None
    x, xin = as_map(x), as_map(xin)
    B, Ho, Wo, C1 = x.t.shape
    B2, H2, W2, C2 = xin.t.shape
    K = conv3.out_channels
    if xin.sub is not None:             # the producer wrote only the pixels this convolution reads (ops.conv1x1_chain(..., sub=))
        if xin.sub != sd[0]:
            raise RuntimeError(f"conv1x1_dual: source sub-sampled by {xin.sub}, convolution stride {sd[0]}")
        sd = (1, 1)
    if B2 != B or (H2 - 1) // sd[0] + 1 != Ho or (W2 - 1) // sd[0] + 1 != Wo or C1 != conv3.in_channels \
            or C2 != ds_conv.in_channels or x.t.dtype != torch.bfloat16 or xin.t.dtype != torch.bfloat16:
        return None
    if not _lib.load().mv_conv1x1_dual_supported(B * Ho * Wo, C1, C2, K, DT[dt]):
        return None
    wcat, shift = _dual_weights(conv3, bn3, ds_conv, ds_bn)
    y = empty((B, Ho, Wo, K), torch.bfloat16)
    _splitk_scratch(B * Ho * Wo, K, C1 + C2)
    _lib.call("mv_conv1x1_dual_fwd", _ptr(x.t), _ptr(xin.t), _ptr(wcat), None, _ptr(shift), _ptr(y), B, Ho, Wo, C1, H2, W2, C2,
              sd[0], K, ACT[act], DT[dt], stream_ptr())
    return Act(y, "map", x.batched)


def conv1x1_dual_available(M: int, conv3, bn3, ds_conv, ds_bn) -> bool:
    """Would ops.conv1x1_dual run for M output pixels of these layers?  (Asked BEFORE a producer decides to write only the pixels
    a strided downsample branch reads.)"""
    if compute_dtype() != "bf16" or not _pointwise(conv3) or _bn_training(bn3) or _bn_training(ds_bn) \
            or conv3.out_channels != ds_conv.out_channels or _lib.get_flag("no_chain_sub"):
        return False
    return bool(_lib.load().mv_conv1x1_dual_supported(M, conv3.in_channels, ds_conv.in_channels, conv3.out_channels, DT["bf16"]))


def chain_rc_available(x: Act, b0, cd, b1, b2) -> bool:
    """Can a three-bottleneck stage run with its first block output un-written (models/classification/resnet.py: _stage_rc)?"""
    if compute_dtype() != "bf16" or _lib.get_flag("no_chain_rc"):
        return False
    convs = (b0.conv3, cd[0], b1.conv1, b1.conv3, b2.conv1)
    if not all(_pointwise(c) for c in convs) or any(_bn_training(b) for b in (b0.bn1, b0.bn2, b0.bn3, cd[1], b1.bn1, b1.bn2, b1.bn3, b2.bn1)):
        return False
    if x.kind != "map" or x.t.dtype != torch.bfloat16:
        return False
    B, H, W, C = x.t.shape
    M, K = B * H * W, b0.conv3.out_channels
    if b0.conv3.in_channels != C or cd[0].in_channels != C or b1.conv3.in_channels != C or cd[0].out_channels != K \
            or b1.conv3.out_channels != K or b1.conv1.in_channels != K or b2.conv1.in_channels != K \
            or tuple(b0.conv2.stride) != (1, 1) or tuple(b1.conv2.stride) != (1, 1) or b0.conv2.out_channels != C or b1.conv2.out_channels != C:
        return False
    lib = _lib.load()
    return bool(lib.mv_conv1x1_dual_chain_supported(M, C, C, K, b1.conv1.out_channels, DT["bf16"])
                and lib.mv_conv1x1_chain_rc_supported(M, C, K, b2.conv1.out_channels, DT["bf16"]))


def conv1x1_dual_chain(x: Act, conv3, bn3, xin: Act, ds_conv, ds_bn, conv1n, bn1n, store_y: bool = True) -> Optional[Act]:
    """relu(bn3(conv3(x)) + ds_bn(ds_conv(xin))) -- a bottleneck whose identity is a pointwise conv of the block input
    (resnet.py:295-303) -- and relu(bn1n(conv1n(.))) of that result, in ONE launch: the two convolutions that add into
    the same output run as one GEMM over the concatenated reduction [x | xin], the BatchNorm scales folded into the bf16
    weight rows.  Returns the block output with the next conv1's result attached (`.pre`), or None if unsupported."""
    dt = compute_dtype()
    if dt != "bf16" or not (_pointwise(conv3) and _pointwise(ds_conv) and _pointwise(conv1n)):
        return None
    if conv3.out_channels != ds_conv.out_channels or conv3.out_channels != conv1n.in_channels:
        return None
    if any(_bn_training(b) for b in (bn3, ds_bn, bn1n)):
        return None
    x, xin = as_map(x), as_map(xin)
    B, H, W, C1 = x.t.shape
    C2, K, N2 = xin.t.shape[-1], conv3.out_channels, conv1n.out_channels
    M = B * H * W
    if tuple(xin.t.shape[:3]) != (B, H, W) or C1 != conv3.in_channels or C2 != ds_conv.in_channels \
            or x.t.dtype != torch.bfloat16 or xin.t.dtype != torch.bfloat16:
        return None
    if not _lib.load().mv_conv1x1_dual_chain_supported(M, C1, C2, K, N2, DT[dt]):
        return None
    if not store_y and not _lib.get_flag("no_chain_rc0") and _lib.load().mv_conv1x1_chain_rc_supported(M, C1, K, N2, DT[dt]):
        wf, tab, _ = chain_rc0_fragments(conv3, bn3, ds_conv, ds_bn, conv1n, bn1n)      # the same function without its output map
        t1 = empty((B, H, W, N2), torch.bfloat16)
        _lib.call("mv_conv1x1_chain_rc0_fwd", _ptr(x.t), _ptr(xin.t), _ptr(wf), _ptr(tab), _ptr(t1), M, C1, K, N2, DT[dt], stream_ptr())
        return Act(t1, "map", x.batched)
    wcat, shift = _dual_weights(conv3, bn3, ds_conv, ds_bn)
    w1, s1, h1 = prep_conv(conv1n, bn1n, "krsc", dt)
    y = empty((B, H, W, K), torch.bfloat16) if store_y else None
    t1 = empty((B, H, W, N2), torch.bfloat16)
    _lib.call("mv_conv1x1_dual_chain_fwd", _ptr(x.t), _ptr(xin.t), _ptr(wcat), None, _ptr(shift), _ptr(y), _ptr(w1), _ptr(s1),
              _ptr(h1), _ptr(t1), M, C1, C2, K, N2, DT[dt], stream_ptr())
    if not store_y:                     # the block output stays un-written (the next boundary recomputes it: ops.conv1x1_chain_rc):
        return Act(t1, "map", x.batched)        # the caller gets the next block's conv1 output itself
    out = Act(y, "map", x.batched)
    out.pre = (conv1n, Act(t1, "map", x.batched))
    return out


def stem_conv_pool(x: Act, conv, bn, act, pool) -> Act:
    """ResNet entry (resnet.py:243-254): conv1 + bn1 + relu + maxpool.  One launch when the library has the fused
    path for this configuration (the 112x112x64 map then never reaches HBM), else conv2d followed by maxpool2d."""
    if _bn_training(bn):
        return maxpool2d(conv2d(x, conv, bn, act), pool.kernel_size, pool.stride, pool.padding)
    dt = compute_dtype()
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    pk, ps, pp = _pair(pool.kernel_size), _pair(pool.stride), _pair(pool.padding)
    fused = (x.kind == "img" and conv.groups == 1 and tuple(conv.dilation) == (1, 1) and pk[0] == pk[1] and ps[0] == ps[1]
             and pp[0] == pp[1] and x.t.dim() == 4 and x.t.shape[1] == conv.in_channels)
    if fused:
        B, C, H, W = x.t.shape
        fused = bool(_lib.load().mv_stem_conv_pool_supported(C, conv.out_channels, kh, kw, sh, sw, ph, pw, pk[0], ps[0], pp[0],
                                                              ACT[act], x.dt, DT[dt], B * C * H * W))
    if not fused:
        return maxpool2d(conv2d(x, conv, bn, act), pool.kernel_size, pool.stride, pool.padding)
    w, scale, shift = prep_conv(conv, bn, "oihw", dt)
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    Po = (Ho + 2 * pp[0] - pk[0]) // ps[0] + 1
    Qo = (Wo + 2 * pp[0] - pk[0]) // ps[0] + 1
    y = empty((B, Po, Qo, conv.out_channels), TORCH_DT[dt])
    _lib.call("mv_stem_conv_pool_fwd", _ptr(x.t), _ptr(w), _ptr(scale), _ptr(shift), _ptr(y), B, C, H, W, conv.out_channels,
              kh, kw, sh, sw, ph, pw, pk[0], ps[0], pp[0], ACT[act], x.dt, DT[dt], stream_ptr())
    return Act(y, "map", x.batched)


def linear(x: Act, lin, act=None, residual: Optional[Act] = None, out_fp32: bool = False) -> Act:
    """Linear over the last (feature) axis of rows: seq [B,N,D], vec [B,D] or map (Linear2d)."""
    dt = compute_dtype()
    if act in UNFUSED_ACTS:
        return eltwise(linear(x, lin, None, residual, out_fp32), act)
    x = as_map(x) if x.kind in ("img", "map") else as_rows(x)
    if x.t.dtype != TORCH_DT[dt]:
        x = cast(x, dt)
    N, K = lin.out_features, lin.in_features
    if x.t.shape[-1] != K:
        raise ValueError(f"Linear expected {K} input features, got {x.t.shape[-1]}")
    M = x.t.numel() // K
    if residual is not None and residual.t.dtype == torch.float32:
        out_fp32 = True                 # fp32 residual stream: add and store in fp32
    odt = torch.float32 if out_fp32 else TORCH_DT[dt]
    y = empty(tuple(x.t.shape[:-1]) + (N,), odt)
    res = None
    if residual is not None:
        if tuple(residual.t.shape) != tuple(y.shape) or residual.t.dtype != odt:
            raise ValueError(f"residual shape/dtype mismatch in linear: {residual} vs {tuple(y.shape)} {odt}")
        res = residual.t
    odc = _lib.F32 if out_fp32 else DT[dt]
    if res is None and dt == "bf16" and _lib.load().mv_fc_stream_supported(M, N, K, DT[dt], odc):
        # few rows, a big weight matrix (AlexNet / VGG classifiers): the layer is bound by streaming W -- fragment-ordered weights,
        # split-K over the CUs, fixed-order reduction (csrc/fc_stream.hip)
        # (decided BEFORE prep_linear: the row-major bf16 copy of a 9216 x 4096 weight would only double the device memory)
        wf = fc_fragments(lin)
        b = prep_f32(lin, "bias", None if lin.bias is None else np.asarray(lin.bias, np.float32).reshape(-1))
        nbytes = int(_lib.load().mv_fc_stream_workspace(M, N, K))
        ws = _fc_workspace(nbytes)
        _lib.call("mv_fc_stream_fwd", _ptr(x.t), _ptr(wf), _ptr(b), _ptr(y), _ptr(ws), nbytes, M, N, K, ACT[act], DT[dt], odc, stream_ptr())
        return Act(y, x.kind, x.batched)
    w, b = prep_linear(lin, dt)
    if dt == "bf16":
        _splitk_scratch(M, N, K)
    _lib.call("mv_linear_fwd", _ptr(x.t), _ptr(w), None, _ptr(b), _ptr(res), _ptr(y), M, N, K, ACT[act],
              DT[dt], odc, stream_ptr())
    return Act(y, x.kind, x.batched)


_FC_WS = {}


def _fc_workspace(nbytes: int) -> torch.Tensor:
    """The split-K workspace of an eager call: ONE per stream instead of a fresh allocation per call (launches on a stream are
    ordered, so consecutive layers share it).  While a forward is being RECORDED every call keeps its own: the lanes of a recording
    become parallel branches of the hipGraph, and two branches must not share scratch memory."""
    from . import _act
    if getattr(_act._tls, "keep", None) is not None:
        return empty(((nbytes + 3) // 4,), torch.float32)
    key = (stream_ptr(), torch.cuda.current_device())
    ws = _FC_WS.get(key)
    if ws is None or ws.numel() * 4 < nbytes:
        ws = torch.empty(((nbytes + 3) // 4,), dtype=torch.float32, device=device())
        _FC_WS[key] = ws
    return ws


_SPLITK_WS = {}


def _splitk_scratch(M: int, N: int, kred: int):
    """Hand the next launch on this stream room for split-K partial sums (include/eqxvision_amd.h: mv_set_scratch) when its shape
    is one the library would split: [M] x [N] output, `kred` reduction elements.  Same ownership rule as `_fc_workspace`: while
    a forward is being recorded every launch keeps its own (lanes become parallel graph branches), an eager call shares one per
    stream.  The first 4096 bytes (the tiles' arrival words) are zeroed once; the kernel leaves them zero."""
    nbytes = int(_lib.load().mv_splitk_scratch_bytes(M, N, kred))
    if not nbytes:
        _lib.call("mv_set_scratch", None, 0, stream_ptr())       # withdraw whatever an earlier entry left pending on this thread
        return
    from . import _act
    if getattr(_act._tls, "keep", None) is not None:
        ws = empty((nbytes,), torch.uint8)
        ws[:4096].zero_()
    else:
        key = (stream_ptr(), torch.cuda.current_device())
        ws = _SPLITK_WS.get(key)
        if ws is None or ws.numel() < nbytes:
            ws = torch.zeros((nbytes,), dtype=torch.uint8, device=device())
            _SPLITK_WS[key] = ws
    _lib.call("mv_set_scratch", _ptr(ws), nbytes, stream_ptr())


def fc_fragments(lin) -> torch.Tensor:
    """Linear weight [N][K] in MFMA fragment order for csrc/fc_stream.hip: [N / 32 tiles][K / 16 steps][64 lanes][8] bf16, lane
    (n = l % 32, h = l / 32) holds W[32 tile + n][16 step + 8 h .. + 7]; N is padded to a multiple of 32 with zero rows (cached)."""
    cache = lin._cache()
    hit = cache.get("fc_frag")
    if hit is None:
        w = torch.from_numpy(np.ascontiguousarray(np.asarray(lin.weight, np.float32))).to(torch.bfloat16)
        N, K = w.shape
        NT = (N + 31) // 32
        if NT * 32 != N:
            w = torch.cat([w, torch.zeros((NT * 32 - N, K), dtype=torch.bfloat16)], dim=0)
        # [NT][32 n][K/16][2 h][8] -> [NT][K/16][2 h][32 n][8]
        hit = w.view(NT, 32, K // 16, 2, 8).permute(0, 2, 3, 1, 4).contiguous().to(device())
        cache["fc_frag"] = hit
    return hit


def linear_head(x: Act, lin) -> Act:
    """The classifier head -> fp32 logits.  With `head_fp32()` (default in bf16 mode) features and weights stay fp32 and the
    product runs on the exact-fp32 MFMA; otherwise the bf16 Linear with an fp32 store."""
    if not head_fp32():
        return linear(x, lin, out_fp32=True)
    if x.t.dtype != torch.float32:
        x = cast(x, "fp32")
    w, b = prep_linear(lin, "fp32")
    N, K = lin.out_features, lin.in_features
    if x.t.shape[-1] != K:
        raise ValueError(f"Linear expected {K} input features, got {x.t.shape[-1]}")
    M = x.t.numel() // K
    y = empty(tuple(x.t.shape[:-1]) + (N,), torch.float32)
    _lib.call("mv_linear_fwd", _ptr(x.t), _ptr(w), None, _ptr(b), None, _ptr(y), M, N, K, _lib.ACT_NONE, _lib.F32, _lib.F32,
              stream_ptr())
    return Act(y, x.kind, x.batched)


def linear_split(x: Act, lin, out_fp32: bool = False, act=None, residual: Optional[Act] = None) -> Act:
    """Linear with split-precision (hi + lo bf16) weights where the library has the path, else the plain Linear."""
    dt = compute_dtype()
    x = as_map(x) if x.kind in ("img", "map") else as_rows(x)
    N, K = lin.out_features, lin.in_features
    M = x.t.numel() // K
    if not split_weights() or act in UNFUSED_ACTS or x.t.dtype != torch.bfloat16 or x.t.shape[-1] != K or \
            not _lib.load().mv_linear_split_supported(M, N, K, DT[dt]):
        return linear(x, lin, act=act, residual=residual, out_fp32=out_fp32)
    if residual is not None and residual.t.dtype == torch.float32:
        out_fp32 = True
    w, b = prep_linear_split(lin)
    odt = torch.float32 if out_fp32 else TORCH_DT[dt]
    y = empty(tuple(x.t.shape[:-1]) + (N,), odt)
    res = None
    if residual is not None:
        if tuple(residual.t.shape) != tuple(y.shape) or residual.t.dtype != odt:
            raise ValueError(f"residual shape/dtype mismatch in linear_split: {residual} vs {tuple(y.shape)} {odt}")
        res = residual.t
    _splitk_scratch(M, N, 2 * K)
    _lib.call("mv_linear_split_fwd", _ptr(x.t), _ptr(w), None, _ptr(b), _ptr(res), _ptr(y), M, N, K, ACT[act], DT[dt],
              _lib.F32 if out_fp32 else DT[dt], stream_ptr())
    return Act(y, x.kind, x.batched)


_SWIN_FUSED_WIDTHS = (96, 192, 384, 768)


def swin_precise(C: int) -> bool:
    """Swin widths outside swin_t / swin_s's (swin_b: 128 / 256 / 512 / 1024) take the un-fused block path; there the bf16 rounding
    of the block Linears' WEIGHTS alone is 1.16e-2 of absolute logit error over 24 blocks (tests/attrib_swin_bf16.py: weights
    bf16 / activations fp32 1.16e-2, weights fp32 / activations bf16 2.2e-3), past north_star's 1e-2 -- so those Linears run with
    split-precision weights (two products per layer; swin_b is not a measured config, swin_t's kernels are untouched)."""
    return split_weights() and C not in _SWIN_FUSED_WIDTHS and not _lib.get_flag("no_swin_precise")


def conv2d_entry_split(x: Act, conv) -> Optional[Act]:
    """The network-entry convolution (raw NCHW image, no norm) with split-precision weights; None if not applicable."""
    dt = compute_dtype()
    if not split_weights() or x.kind != "img" or conv.groups != 1 or tuple(conv.dilation) != (1, 1) or \
            conv.in_channels > STEM_MAX_CIN or x.t.shape[1] != conv.in_channels:
        return None
    B, C, H, W = x.t.shape
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    K = conv.out_channels
    hi, lo, b = prep_conv_split(conv)
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    y = empty((B, Ho, Wo, K), TORCH_DT[dt])
    try:
        _lib.call("mv_conv2d_nchw_split_fwd", _ptr(x.t), _ptr(hi), _ptr(lo), None, _ptr(b), _ptr(y), B, C, H, W, K, kh, kw, sh, sw,
                  ph, pw, _lib.ACT_NONE, x.dt, DT[dt], stream_ptr())
    except _lib.MVError:
        return None
    return Act(y, "map", x.batched)


def patch4_ln(x: Act, conv, ln) -> Optional[Act]:
    """LayerNorm2d(Conv2d(3, K, 4, 4)(x)) -- Swin's patch embedding and its norm (swin.py:705-711) -- in one launch from the raw
    NCHW image to the fp32 NHWC residual stream, split-precision weights; None when the library has no such path."""
    if compute_dtype() != "bf16" or x.kind != "img" or x.t.dtype != torch.float32 or conv.groups != 1 or ln.weight is None or ln.bias is None:
        return None
    B, C, H, W = x.t.shape
    K = conv.out_channels
    if tuple(conv.kernel_size) != (4, 4) or tuple(conv.stride) != (4, 4) or tuple(conv.padding) != (0, 0) or tuple(conv.dilation) != (1, 1) \
            or conv.in_channels != C or int(np.prod(ln.shape)) != K:
        return None
    if not _lib.load().mv_patch4_ln_supported(C, H, W, K, _lib.F32):
        return None
    hi, lo, b = prep_conv_split(conv)
    if not split_weights():
        lo = None
    y = empty((B, H // 4, W // 4, K), torch.float32)
    _lib.call("mv_patch4_ln_fwd", _ptr(x.t), _ptr(hi), _ptr(lo), _ptr(b), _ptr(prep_f32(ln, "weight", ln.weight)),
              _ptr(prep_f32(ln, "bias", ln.bias)), _ptr(y), B, C, H, W, K, float(ln.eps), _lib.F32, stream_ptr())
    return Act(y, "map", x.batched)


def patch_embed_tokens(x: Act, conv, cls: Optional[torch.Tensor], pos: Optional[torch.Tensor], n_extra: int, out_fp32: bool = False) -> Act:
    """PatchEmbed conv (k = s = patch) straight from the NCHW image into token rows
    [B, n_extra + P, D]; with `pos` the position embedding is added in the epilogue and row 0 gets
    cls + pos[0] (vit.py:268-269).  `out_fp32` (bf16 mode with the fp32 residual stream): the rows are written in fp32 where the
    library has that epilogue, else in the compute dtype (the caller casts)."""
    dt = compute_dtype()
    if x.kind != "img":
        raise ValueError("patch_embed expects a raw (C,H,W) image")
    B, C, H, W = x.t.shape
    kh, kw = conv.kernel_size
    sh, sw = conv.stride
    ph, pw = conv.padding
    K = conv.out_channels
    w, scale, shift = prep_conv(conv, None, "oihw", dt)
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    T = n_extra + Ho * Wo
    if out_fp32 and dt == "bf16" and _lib.load().mv_conv2d_nchw_f32out_supported(C, H, W, K, kh, kw, sh, sw, ph, pw, x.dt):
        y = empty((B, T, K), torch.float32)
        _lib.call("mv_conv2d_nchw_f32out_fwd", _ptr(x.t), _ptr(w), _ptr(scale), _ptr(shift), _ptr(y),
                  B, C, H, W, K, kh, kw, sh, sw, ph, pw, _lib.ACT_NONE, x.dt, T, n_extra, _ptr(pos), stream_ptr())
        if n_extra:
            _lib.call("mv_vit_cls_pos_fwd", _ptr(cls), _ptr(pos), _ptr(y), B, T, K, _lib.F32, stream_ptr())
        return Act(y, "seq", x.batched)
    y = empty((B, T, K), TORCH_DT[dt])
    _lib.call("mv_conv2d_nchw_fwd", _ptr(x.t), _ptr(w), _ptr(scale), _ptr(shift), _ptr(y),
              B, C, H, W, K, kh, kw, sh, sw, ph, pw, _lib.ACT_NONE, x.dt, DT[dt], T, n_extra, _ptr(pos), stream_ptr())
    if n_extra:
        _lib.call("mv_vit_cls_pos_fwd", _ptr(cls), _ptr(pos), _ptr(y), B, T, K, DT[dt], stream_ptr())
    return Act(y, "seq", x.batched)


# ------------------------------------------------------------------ normalisation / pooling
def layernorm(x: Act, ln, out_fp32: bool = False) -> Act:
    """Rows of the last axis; reads the input in its own dtype (bf16, or the fp32 residual stream) and
    writes the compute dtype (what the next GEMM consumes) unless `out_fp32`."""
    if x.kind == "img":
        x = as_map(x)
    dt = compute_dtype()
    C = x.t.shape[-1]
    if int(np.prod(ln.shape)) != C:
        raise ValueError(f"LayerNorm over {ln.shape} applied to rows of {C}")
    g = prep_f32(ln, "weight", ln.weight)
    b = prep_f32(ln, "bias", ln.bias)
    odt = torch.float32 if out_fp32 else TORCH_DT[dt]
    y = empty(tuple(x.t.shape), odt)
    _lib.call("mv_layernorm_fwd", _ptr(x.t), _ptr(g), _ptr(b), _ptr(y), x.t.numel() // C, C, 0, float(ln.eps),
              x.dt, _lib.F32 if out_fp32 else DT[dt], stream_ptr())
    return Act(y, x.kind, x.batched)


def _ln_folded(lin, ln):
    """(W . diag(gamma) as bf16, b + W . beta as fp32) on the device, cached on the Linear."""
    cache = lin._cache()
    key = ("ln_fold", id(ln.weight), id(ln.bias))
    hit = cache.get(key)
    if hit is None:
        w = np.asarray(lin.weight, np.float32)
        g, b = np.asarray(ln.weight, np.float32).reshape(-1), np.asarray(ln.bias, np.float32).reshape(-1)
        b0 = np.zeros(w.shape[0], np.float32) if lin.bias is None else np.asarray(lin.bias, np.float32).reshape(-1)
        hit = (_dev(w * g[None, :], torch.bfloat16), _dev(b0 + w @ b, torch.float32), ln)      # ln: keeps the ids alive
        cache[key] = hit
    return hit


def _ln_folded_cs(lin, ln):
    """For mv_linear_lnin_fwd: (W' = bf16(W . diag(gamma)), colsum[n] = sum_k W'[n][k] of the ROUNDED values, b' = b + W . beta),
    on the device, cached on the Linear."""
    cache = lin._cache()
    key = ("ln_fold_cs", id(ln.weight), id(ln.bias))
    hit = cache.get(key)
    if hit is None:
        w = np.asarray(lin.weight, np.float32)
        g, b = np.asarray(ln.weight, np.float32).reshape(-1), np.asarray(ln.bias, np.float32).reshape(-1)
        b0 = np.zeros(w.shape[0], np.float32) if lin.bias is None else np.asarray(lin.bias, np.float32).reshape(-1)
        wf = torch.from_numpy(np.ascontiguousarray(w * g[None, :])).to(torch.bfloat16)
        cs = wf.to(torch.float64).sum(dim=1).to(torch.float32).numpy()
        bf = (b0.astype(np.float64) + w.astype(np.float64) @ b.astype(np.float64)).astype(np.float32)
        hit = (wf.to(device()), _dev(cs, torch.float32), _dev(bf, torch.float32), ln)      # ln: keeps the ids alive
        cache[key] = hit
    return hit


def _ln_foldable(ln, C: int) -> bool:
    from . import nn
    return (type(ln) is nn.LayerNorm and ln.weight is not None and ln.bias is not None and int(np.prod(ln.shape)) == C)


def linear_lnout_available(M: int, lin) -> bool:
    return (compute_dtype() == "bf16" and residual_fp32()
            and bool(_lib.load().mv_linear_lnout_supported(M, lin.out_features, lin.in_features, _lib.BF16)))


def linear_lnin_available(M: int, ln, lin, tokens: int = 0, dh: int = 0) -> bool:
    return (compute_dtype() == "bf16" and _ln_foldable(ln, lin.in_features)
            and bool(_lib.load().mv_linear_lnin_supported(M, lin.out_features, lin.in_features, tokens, dh, _lib.BF16)))


def is_split_stream(x: Act) -> bool:
    """The residual stream as two bf16 planes: `x.t` = hi = bf16(y), `x.ln` = (lo = bf16(y - hi), row-statistics pieces)."""
    return x.ln is not None and x.t.dtype == torch.bfloat16


def stream_f32(x: Act) -> Act:
    """Split stream -> fp32 rows (hi + lo), for a consumer that has no folded form (three element-wise launches: not the hot path)."""
    if not is_split_stream(x):
        return x
    hi = Act(x.t, x.kind, x.batched)
    lo = Act(x.ln[0], x.kind, x.batched)
    return add(cast(hi, "fp32"), cast(lo, "fp32"))


def linear_lnout(x: Act, lin, residual: Act, out_split: bool = True) -> Act:
    """residual + lin(x) on the residual STREAM of a block whose LayerNorms are folded into the GEMM epilogues (header:
    mv_linear_lnout_fwd).  `residual`: fp32 rows or a split stream; result: a split stream (`Act.t` = the high plane = the operand of
    the LayerNorm + Linear pair behind it, `Act.ln` = (low plane, statistics pieces) -> `linear_lnin`), or fp32 rows
    (`out_split=False`, from a split residual).  Ask `linear_lnout_available` first."""
    x = as_rows(x)
    if x.t.dtype != torch.bfloat16:
        x = cast(x, "bf16")
    N, K = lin.out_features, lin.in_features
    M = x.t.numel() // K
    split_in = is_split_stream(residual)
    if tuple(residual.t.shape) != tuple(x.t.shape[:-1]) + (N,) or not (split_in or residual.t.dtype == torch.float32):
        raise ValueError(f"linear_lnout: residual {residual} does not match {tuple(x.t.shape[:-1]) + (N,)} (fp32 rows or a split stream)")
    if not out_split and not split_in:
        raise ValueError("linear_lnout: fp32 rows in and out is ops.linear")
    w, b = prep_linear(lin, "bf16")
    res_lo = residual.ln[0] if split_in else None
    if out_split:
        y = empty(tuple(residual.t.shape), torch.bfloat16)
        y_lo = empty(tuple(residual.t.shape), torch.bfloat16)
        st = empty(((N + 255) // 256, M, 2), torch.float32)
    else:
        y, y_lo, st = empty(tuple(residual.t.shape), torch.float32), None, None
    _lib.call("mv_linear_lnout_fwd", _ptr(x.t), _ptr(w), _ptr(b), _ptr(residual.t), _ptr(res_lo), _ptr(y), _ptr(y_lo), _ptr(st), M, N, K,
              _lib.BF16, stream_ptr())
    out = Act(y, x.kind, x.batched)
    if out_split:
        out.ln = (y_lo, st)
    return out


def linear_lnin(x: Act, ln, lin, act=None, heads: int = 0):
    """lin(ln(x)) (+ activation) for a split stream x (header: mv_linear_lnin_fwd).  heads > 0: the qkv projection, written
    head-major [B, 3*heads, N, dh] (as `qkv_attention` does).  Ask `linear_lnin_available` first."""
    hi, st = x.t, x.ln[1]
    N, K = lin.out_features, lin.in_features
    M = hi.numel() // K
    w, cs, b = _ln_folded_cs(lin, ln)[:3]
    if heads:
        B, T, D = hi.shape
        y = empty((B, 3 * heads, T, D // heads), torch.bfloat16)
        tok, dh = T, D // heads
    else:
        y = empty(tuple(hi.shape[:-1]) + (N,), torch.bfloat16)
        tok, dh = 0, 0
    _lib.call("mv_linear_lnin_fwd", _ptr(hi), _ptr(st), _ptr(w), _ptr(cs), _ptr(b), _ptr(y), M, N, K, float(ln.eps), ACT[act], tok, dh,
              _lib.BF16, stream_ptr())
    return y if heads else Act(y, x.kind, x.batched)


def ln_linear(x: Act, ln, lin, act=None) -> Act:
    """lin(ln(x)) (+ activation): ONE launch where the library folds the LayerNorm into the Linear's operand path (short rows:
    Swin stages 0-1), else LayerNorm launch + Linear launch."""
    from . import nn
    C = lin.in_features
    ok = (compute_dtype() == "bf16" and x.kind in ("map", "seq") and x.t.dtype in (torch.float32, torch.bfloat16)
          and isinstance(ln, nn.LayerNorm) and ln.weight is not None and ln.bias is not None
          and x.t.shape[-1] == C and int(np.prod(ln.shape)) == C)
    if ok:
        M = x.t.numel() // C
        xdt = _lib.F32 if x.t.dtype == torch.float32 else _lib.BF16
        ok = bool(_lib.load().mv_ln_linear_supported(M, lin.out_features, C, xdt, _lib.BF16))
    if not ok:
        return linear(layernorm(x, ln), lin, act=act)
    w, b = _ln_folded(lin, ln)[:2]
    y = empty(tuple(x.t.shape[:-1]) + (lin.out_features,), torch.bfloat16)
    _lib.call("mv_ln_linear_fwd", _ptr(x.t), _ptr(w), _ptr(b), _ptr(y), M, lin.out_features, C, float(ln.eps), ACT[act], xdt,
              _lib.BF16, stream_ptr())
    return Act(y, x.kind, x.batched)


def ln_mlp(x: Act, ln, mlp) -> Optional[Act]:
    """x + mlp(ln(x)) in ONE launch where the library has the fused kernel (narrow rows: Swin stage 0), else None.
    The LayerNorm affine is folded into fc1 on the host (fp32, then one bf16 rounding of the product)."""
    from . import nn
    if compute_dtype() != "bf16" or x.kind not in ("map", "seq") or x.t.dtype not in (torch.float32, torch.bfloat16):
        return None
    fc1, fc2 = mlp.fc1, mlp.fc2
    if not (isinstance(fc1, nn.Linear) and isinstance(fc2, nn.Linear) and isinstance(ln, nn.LayerNorm)):
        return None
    if any(not (d.inference or d.p == 0.0) for d in (mlp.drop1, mlp.drop2) if isinstance(d, nn.Dropout)):
        return None
    if nn.act_name(mlp.act) != "gelu" or fc1.bias is None or fc2.bias is None or ln.weight is None or ln.bias is None:
        return None
    C, Hd = fc1.in_features, fc1.out_features
    if x.t.shape[-1] != C or fc2.in_features != Hd or fc2.out_features != C or int(np.prod(ln.shape)) != C:
        return None
    M = x.t.numel() // C
    xdt = _lib.F32 if x.t.dtype == torch.float32 else _lib.BF16
    if not _lib.load().mv_ln_mlp_supported(M, C, Hd, xdt):
        if _lib.load().mv_ln_mlp_stream_supported(M, C, Hd, xdt):
            return _ln_mlp_stream(x, ln, mlp, M, C, Hd, xdt)
        return None
    cache = mlp._cache()
    key = ("ln_mlp", id(ln.weight), id(ln.bias))
    hit = cache.get(key)
    if hit is None:
        w1 = np.asarray(fc1.weight, np.float32)
        g, b = np.asarray(ln.weight, np.float32).reshape(-1), np.asarray(ln.bias, np.float32).reshape(-1)
        hit = (_dev(w1 * g[None, :], torch.bfloat16),
               _dev(np.asarray(fc1.bias, np.float32).reshape(-1) + w1 @ b, torch.float32),
               _dev(np.asarray(fc2.weight, np.float32), torch.bfloat16),
               _dev(np.asarray(fc2.bias, np.float32).reshape(-1), torch.float32), ln)        # ln: keeps the ids alive
        cache[key] = hit
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_ln_mlp_fwd", _ptr(x.t), _ptr(hit[0]), _ptr(hit[1]), _ptr(hit[2]), _ptr(hit[3]), _ptr(y), M, C, Hd,
              float(ln.eps), xdt, stream_ptr())
    return Act(y, x.kind, x.batched)


def ln_mlp_fragments(w1: np.ndarray, w2: np.ndarray):
    """fc1 [hidden][C] / fc2 [C][hidden] -> the fragment order of mv_ln_mlp_stream_fwd (header)."""
    Hd, C = w1.shape
    w1f = w1.reshape(Hd // 256, 8, 32, C // 16, 2, 8).transpose(0, 1, 3, 4, 2, 5)            # (chunk, wave, j, h, m, e)
    w2f = w2.reshape(C // 32, 32, Hd // 256, 16, 2, 8).transpose(2, 0, 3, 4, 1, 5)           # (chunk, tile, j, h, m, e)
    return np.ascontiguousarray(w1f), np.ascontiguousarray(w2f)


def _ln_mlp_stream(x: Act, ln, mlp, M, C, Hd, xdt) -> Act:
    """ln_mlp's wide-row variant: weights streamed from L2 in fragment order (LayerNorm affine folded into fc1 on the host)."""
    fc1, fc2 = mlp.fc1, mlp.fc2
    cache = mlp._cache()
    key = ("ln_mlp_stream", id(ln.weight), id(ln.bias))
    hit = cache.get(key)
    if hit is None:
        w1 = np.asarray(fc1.weight, np.float32)
        g, b = np.asarray(ln.weight, np.float32).reshape(-1), np.asarray(ln.bias, np.float32).reshape(-1)
        w1f, w2f = ln_mlp_fragments(w1 * g[None, :], np.asarray(fc2.weight, np.float32))
        hit = (_dev(w1f, torch.bfloat16), _dev(np.asarray(fc1.bias, np.float32).reshape(-1) + w1 @ b, torch.float32),
               _dev(w2f, torch.bfloat16), _dev(np.asarray(fc2.bias, np.float32).reshape(-1), torch.float32), ln)
        cache[key] = hit
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_ln_mlp_stream_fwd", _ptr(x.t), _ptr(hit[0]), _ptr(hit[1]), _ptr(hit[2]), _ptr(hit[3]), _ptr(y), M, C, Hd,
              float(ln.eps), xdt, stream_ptr())
    return Act(y, x.kind, x.batched)


def cast(x: Act, dtype: str) -> Act:
    if x.t.dtype == TORCH_DT[dtype]:
        return x
    y = empty(tuple(x.t.shape), TORCH_DT[dtype])
    _lib.call("mv_cast", _ptr(x.t), _ptr(y), x.t.numel(), x.dt, DT[dtype], stream_ptr())
    return Act(y, x.kind, x.batched)


def layernorm_first_row(x: Act, ln, out_fp32: bool = False) -> Act:
    """LayerNorm of row 0 of every sample of a seq [B,N,D] -> vec [B,D] (vit.py:272-273: only
    `x[0]` of the normalised tokens is used), via the kernel's row-stride argument."""
    dt = compute_dtype()
    B, N, D = x.t.shape
    g = prep_f32(ln, "weight", ln.weight)
    b = prep_f32(ln, "bias", ln.bias)
    y = empty((B, D), torch.float32 if out_fp32 else TORCH_DT[dt])
    _lib.call("mv_layernorm_fwd", _ptr(x.t), _ptr(g), _ptr(b), _ptr(y), B, D, N * D, float(ln.eps),
              x.dt, _lib.F32 if out_fp32 else DT[dt], stream_ptr())
    return Act(y, "vec", x.batched)


def batchnorm(x: Act, bn, act=None) -> Act:
    """Stand-alone BatchNorm (unfused call sites): per-channel affine with the running statistics -- after updating them from
    this batch when the module is in training mode (reference semantics)."""
    if x.kind == "seq":
        # eqx.experimental.BatchNorm treats AXIS 0 of the single-sample array as channels; a (tokens, features) array would be
        # normalised per token there, per feature here -- refuse instead of returning different numbers (round-1 advice)
        raise NotImplementedError("BatchNorm on a (tokens, features) array: the reference normalises axis 0; not on the hot path")
    x = as_map(x) if x.kind in ("img", "map") else as_rows(x)
    if _bn_training(bn):
        sc, sh = bn_train_update(bn, x)
        return _bn_affine(x, sc, sh, act)
    cache = bn._cache()
    hit = cache.get(("fold", bn.state_index.version))
    if hit is None:
        s, h = bn_fold(bn)
        hit = (_dev(s, torch.float32), _dev(h, torch.float32))
        cache[("fold", bn.state_index.version)] = hit
    C = x.t.shape[-1]
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_channel_affine_fwd", _ptr(x.t), _ptr(hit[0]), _ptr(hit[1]), _ptr(y), x.t.numel() // C, C,
              ACT[act], x.dt, stream_ptr())
    return Act(y, x.kind, x.batched)


def maxpool2d(x: Act, kernel_size, stride, padding) -> Act:
    x = as_map(x)
    B, H, W, C = x.t.shape
    kh, kw = _pair(kernel_size)
    sh, sw = _pair(stride)
    ph, pw = _pair(padding)
    Ho = (H + 2 * ph - kh) // sh + 1
    Wo = (W + 2 * pw - kw) // sw + 1
    y = empty((B, Ho, Wo, C), x.t.dtype)
    _lib.call("mv_maxpool2d_nhwc_fwd", _ptr(x.t), _ptr(y), B, H, W, C, kh, kw, sh, sw, ph, pw, x.dt, stream_ptr())
    return Act(y, "map", x.batched)


def adaptive_avgpool2d(x: Act, target, out_fp32: bool = False) -> Act:
    x = as_map(x)
    B, H, W, C = x.t.shape
    oh, ow = _pair(target)
    if oh == H and ow == W and not (out_fp32 and x.t.dtype != torch.float32):
        return x
    odt = torch.float32 if out_fp32 else x.t.dtype
    y = empty((B, oh, ow, C), odt)
    _lib.call("mv_adaptive_avgpool2d_nhwc_fwd", _ptr(x.t), _ptr(y), B, H, W, C, oh, ow, x.dt,
              _lib.F32 if odt == torch.float32 else _lib.BF16, stream_ptr())
    return Act(y, "map", x.batched)


# ------------------------------------------------------------------ attention
def _mha_call(qkv_t, head_major: bool, out, probs, B, N, heads, dh, scale, dt, drop):
    """The attention core; `drop` = (p, keys) runs the reference's live attention dropout inside the kernel (vit.py:71)."""
    if drop is None:
        _lib.call("mv_mha_heads_fwd" if head_major else "mv_mha_fwd", _ptr(qkv_t), _ptr(out), _ptr(probs), B, N, heads, dh,
                  float(scale), dt, stream_ptr())
        return
    p, key = drop
    if not 0.0 < float(p) < 1.0:
        raise NotImplementedError(f"attention Dropout(p={p})")
    keys = _keys_dev(key, B)
    _lib.call("mv_mha_dropout_fwd", _ptr(qkv_t), 1 if head_major else 0, _ptr(out), _ptr(probs), _ptr(keys), float(1.0 - p),
              B, N, heads, dh, float(scale), dt, stream_ptr())


def mha(qkv: Act, heads: int, scale: float, need_probs: bool, drop=None):
    """qkv rows [B,N,3D] -> ([B,N,D], probs fp32 [B,heads,N,N] or None)   (vit.py:65-73)."""
    qkv = cast(qkv, compute_dtype())
    B, N, D3 = qkv.t.shape
    D = D3 // 3
    dh = D // heads
    out = empty((B, N, D), qkv.t.dtype)
    probs = empty((B, heads, N, N), torch.float32) if need_probs else None
    _mha_call(qkv.t, False, out, probs, B, N, heads, dh, scale, qkv.dt, drop)
    return Act(out, "seq", qkv.batched), probs


def qkv_attention_ln(x: Act, ln, lin, heads: int, scale: float):
    """`qkv_attention(ln(x), lin, ...)` for a split stream x (inference, no probabilities): the LayerNorm is folded into the qkv
    projection's epilogue (vit.py:142 norm1 -> :64 qkv)."""
    B, N, D = x.t.shape
    qkv = linear_lnin(x, ln, lin, heads=heads)
    out = empty((B, N, D), torch.bfloat16)
    _mha_call(qkv, True, out, None, B, N, heads, D // heads, scale, _lib.BF16, None)
    return Act(out, "seq", x.batched)


def qkv_attention(x: Act, lin, heads: int, scale: float, need_probs: bool, drop=None):
    """qkv Linear + attention core of `_VitAttention` (vit.py:64-73): rows [B,N,D] -> ([B,N,D], probs or None).
    When the library has the path for this shape the projection writes q/k/v head-major
    ([B, 3*heads, N, dh]: what the reference's reshape + transpose produce) and the attention kernel reads
    contiguous heads; otherwise Linear -> [B,N,3D] -> strided attention.  Both are the HIP path.
    `drop` = (p, per-sample keys): training-mode attention dropout, applied to the probabilities inside the kernel."""
    dt = compute_dtype()
    x = as_rows(x)
    if x.t.dtype != TORCH_DT[dt]:
        x = cast(x, dt)
    B, N, D = x.t.shape
    if lin.in_features != D or lin.out_features != 3 * D or D % heads:
        raise ValueError(f"qkv projection {lin.in_features}->{lin.out_features} does not fit rows of {D} / {heads} heads")
    dh = D // heads
    if not _lib.load().mv_linear_heads_supported(B * N, 3 * D, D, N, dh, DT[dt]):
        return mha(linear(x, lin), heads, scale, need_probs, drop)
    w, b = prep_linear(lin, dt)
    qkv = empty((B, 3 * heads, N, dh), TORCH_DT[dt])
    _lib.call("mv_linear_heads_fwd", _ptr(x.t), _ptr(w), None, _ptr(b), _ptr(qkv), B * N, 3 * D, D, N, dh, DT[dt],
              stream_ptr())
    out = empty((B, N, D), TORCH_DT[dt])
    probs = empty((B, heads, N, N), torch.float32) if need_probs else None
    _mha_call(qkv, True, out, probs, B, N, heads, dh, scale, DT[dt], drop)
    return Act(out, "seq", x.batched), probs


def swin_rel_bias(attn) -> torch.Tensor:
    """The relative-position bias [heads][n][n] of a _ShiftedWindowAttention on the device: table[index] gathered on the host
    (swin.py:34-43), cached on the module."""
    cache = attn._cache()
    b = cache.get("bias")
    if b is None:
        b = _dev(attn.get_relative_position_bias(), torch.float32)
        cache["bias"] = b
    return b


def swin_window_attention(qkv: Act, bias: torch.Tensor, heads: int, window, shift, drop=None) -> Act:
    """`drop` = (p, per-sample keys): the reference's `_func_dropout(attn, attention_dropout, key)` (swin.py:227) inside the kernel."""
    B, Hf, Wf, C3 = qkv.t.shape
    C = C3 // 3
    out = empty((B, Hf, Wf, C), qkv.t.dtype)
    if drop is None:
        _lib.call("mv_swin_window_attn_fwd", _ptr(qkv.t), _ptr(bias), _ptr(out), B, Hf, Wf, C, heads,
                  int(window[0]), int(window[1]), int(shift[0]), int(shift[1]), qkv.dt, stream_ptr())
    else:
        p, key = drop
        if not 0.0 < float(p) < 1.0:
            raise NotImplementedError(f"Swin attention_dropout = {p}")
        _lib.call("mv_swin_window_attn_dropout_fwd", _ptr(qkv.t), _ptr(bias), _ptr(out), _ptr(_keys_dev(key, B)), float(1.0 - p),
                  B, Hf, Wf, C, heads, int(window[0]), int(window[1]), int(shift[0]), int(shift[1]), qkv.dt, stream_ptr())
    return Act(out, "map", qkv.batched)


def swin_block_attn_fragments(wqkv: np.ndarray, bqkv: np.ndarray, wp: np.ndarray, bias: np.ndarray):
    """qkv weight [3C][C] (LayerNorm already folded) / bias [3C], proj weight [C][C], relative-position bias [heads][n][n] ->
    the operand layouts of mv_swin_block_attn_fwd (header), or None for a width the kernel is not instantiated for (swin_b's
    128 / 256 / 512 / 1024: the caller takes the un-fused path)."""
    C = wp.shape[0]
    heads = C // 32
    if C not in (384, 192, 96):                           # the widths the kernel is instantiated for (NWIN / NWV templates)
        return None
    HG = {384: 4, 192: 2, 96: 1}[C]                       # heads per group (x windows per workgroup = 4)
    G, GT = heads // HG, 3 * HG                            # always 3 groups; tiles per group: q heads, k heads, v heads
    rows = np.array([[(t // HG) * C + 32 * (HG * g + t % HG) for t in range(GT)] for g in range(G)])              # [G][GT]
    w = wqkv[(rows[:, :, None] + np.arange(32)[None, None, :])]                                                   # [G][GT][32][C]
    wf = w.reshape(G, GT, 32, C // 16, 2, 8).transpose(0, 1, 3, 4, 2, 5)                                           # (g, t, j, h, m, e)
    bf_ = bqkv[(rows[:, :, None] + np.arange(32)[None, None, :])]                                                  # [G][GT][32]
    wpf = wp.reshape(C // 32, 32, C // 16, 2, 8).transpose(0, 2, 3, 1, 4)                                          # (t, j, h, m, e)
    n = bias.shape[1]
    b64 = np.zeros((heads, 64, 64), np.float32)
    b64[:, :, n:] = -1e30
    b64[:, :n, :n] = bias
    return (np.ascontiguousarray(wf, np.float32), np.ascontiguousarray(bf_, np.float32), np.ascontiguousarray(wpf, np.float32), b64)


def swin_block_attention(x: Act, norm, attn) -> Optional[Act]:
    """x + attn(norm(x)) of a Swin block (swin.py:572-578, first line) in ONE launch where the library has the fused per-window
    kernel (stage 2 of swin_t / swin_s), else None."""
    from . import nn
    if compute_dtype() != "bf16" or x.kind != "map" or x.t.dtype != torch.float32 or not isinstance(norm, nn.LayerNorm):
        return None
    if norm.weight is None or norm.bias is None or attn.qkv.bias is None or attn.proj.bias is None:
        return None
    B, Hf, Wf, C = x.t.shape
    ws, sh = attn.window_size, list(attn.shift_size)
    if ws[0] >= Hf:                                                   # reference :116-120: no shift when the window covers the map
        sh[0] = 0
    if ws[1] >= Wf:
        sh[1] = 0
    if attn.qkv.in_features != C or not _lib.load().mv_swin_block_attn_supported(Hf, Wf, C, attn.num_heads, ws[0], ws[1], _lib.F32):
        return None
    cache = attn._cache()
    key = ("block_attn", id(norm.weight), id(norm.bias))
    hit = cache.get(key)
    if hit is None:
        wq = np.asarray(attn.qkv.weight, np.float32)
        g, b = np.asarray(norm.weight, np.float32).reshape(-1), np.asarray(norm.bias, np.float32).reshape(-1)
        frags = swin_block_attn_fragments(wq * g[None, :], np.asarray(attn.qkv.bias, np.float32).reshape(-1) + wq @ b,
                                          np.asarray(attn.proj.weight, np.float32), attn.get_relative_position_bias())
        if frags is None:
            return None
        wf, bq, wpf, b64 = frags
        hit = (_dev(wf, torch.bfloat16), _dev(bq, torch.float32), _dev(wpf, torch.bfloat16),
               _dev(np.asarray(attn.proj.bias, np.float32).reshape(-1), torch.float32), _dev(b64, torch.float32), norm)
        cache[key] = hit
    y = empty(tuple(x.t.shape), torch.float32)
    _lib.call("mv_swin_block_attn_fwd", _ptr(x.t), _ptr(hit[0]), _ptr(hit[1]), _ptr(hit[2]), _ptr(hit[3]), _ptr(hit[4]), _ptr(y),
              B, Hf, Wf, C, attn.num_heads, ws[0], ws[1], sh[0], sh[1], float(norm.eps), _lib.F32, stream_ptr())
    return Act(y, "map", x.batched)


def dropout_windows(x: Act, p: float, key, window, shift) -> Act:
    """`_func_dropout(x, dropout, key)` on the projection output of `_shifted_window_attention` (swin.py:233): the mask is drawn
    for the (num_windows, tokens, C) layout the reference holds there; x is the NHWC map of the same values."""
    x = as_map(x)
    if not 0.0 < float(p) < 1.0:
        raise NotImplementedError(f"Swin dropout = {p}")
    B, Hf, Wf, C = x.t.shape
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_dropout_windows_fwd", _ptr(x.t), _ptr(_keys_dev(key, B)), _ptr(y), B, Hf, Wf, C, int(window[0]), int(window[1]),
              int(shift[0]), int(shift[1]), float(1.0 - p), x.dt, stream_ptr())
    return Act(y, "map", x.batched)


def resize_bilinear(x: Act, size, final: bool = False) -> Act:
    """jax.image.resize(x, (C,) + size, "bilinear") of a feature map (up-sampling).  `final`: the result is what the caller of the
    model receives -> written directly as fp32 NCHW; otherwise an NHWC map in the compute dtype for further layers."""
    x = as_map(x)
    B, h, w, C = x.t.shape
    H, W = int(size[0]), int(size[1])
    if final:
        y = empty((B, C, H, W), torch.float32)
        _lib.call("mv_resize_bilinear_nhwc_fwd", _ptr(x.t), _ptr(y), B, h, w, C, H, W, x.dt, _lib.F32, 1, stream_ptr())
        return Act(y, "img", x.batched)
    if (H, W) == (h, w):
        return x
    y = empty((B, H, W, C), x.t.dtype)
    _lib.call("mv_resize_bilinear_nhwc_fwd", _ptr(x.t), _ptr(y), B, h, w, C, H, W, x.dt, x.dt, 0, stream_ptr())
    return Act(y, "map", x.batched)


def concat_channels(xs) -> Act:
    """jnp.concatenate(maps, axis=0) of (C,H,W) samples == channel concatenation of NHWC maps (deeplabv3.py:132-136)."""
    maps = [as_map(x) for x in xs]
    B, H, W, _ = maps[0].t.shape
    dt = maps[0].t.dtype
    for m in maps:
        if tuple(m.t.shape[:3]) != (B, H, W) or m.t.dtype != dt:
            raise ValueError(f"concat_channels: mismatched maps {[tuple(m.t.shape) for m in maps]}")
    ctot = sum(m.t.shape[3] for m in maps)
    y = empty((B, H, W, ctot), dt)
    es = y.element_size()
    off = 0
    for m in maps:
        c = m.t.shape[3]
        _lib.call("mv_copy_rows", _ptr(m.t), y.data_ptr() + off * es, B * H * W, c * es, c * es, ctot * es, stream_ptr())
        off += c
    return Act(y, "map", maps[0].batched)


def se_scale(x: Act, fc1, fc2, act1, act2) -> Optional[Act]:
    """SqueezeExcitation's scale vector [B, C] (layers/squeeze.py:47-60: mean -> fc1 -> activation -> fc2 -> scale activation) in
    ONE launch, or None when the library has no fused path (caller: avgpool + two convolutions + activations)."""
    x = as_map(x)
    B, H, W, C = x.t.shape
    S = fc1.out_channels
    plain = all(tuple(c.kernel_size) == (1, 1) and tuple(c.stride) == (1, 1) and tuple(c.padding) == (0, 0) and c.groups == 1
                for c in (fc1, fc2))
    if (compute_dtype() != "bf16" or x.t.dtype != torch.bfloat16 or not plain or fc1.in_channels != C or fc2.in_channels != S
            or fc2.out_channels != C or act1 not in ACT or act2 not in ACT or not _lib.load().mv_se_scale_supported(C, S, _lib.BF16)):
        return None
    w1, _, b1 = prep_conv(fc1, None, "krsc", "bf16")
    cache = fc2._cache()
    hit = cache.get("se_w2t")
    if hit is None:                     # [S][C]: the kernel's threads read consecutive channels of a row
        w = np.asarray(fc2.weight, np.float32).reshape(C, S)
        hit = (_dev(np.ascontiguousarray(w.T), torch.bfloat16),
               None if fc2.bias is None else _dev(np.asarray(fc2.bias, np.float32).reshape(-1), torch.float32))
        cache["se_w2t"] = hit
    w2, b2 = hit
    s = empty((B, C), torch.bfloat16)
    _lib.call("mv_se_scale_fwd", _ptr(x.t), _ptr(w1), _ptr(b1), _ptr(w2), _ptr(b2), _ptr(s), B, H * W, C, S, ACT[act1], ACT[act2],
              _lib.BF16, stream_ptr())
    return Act(s, "vec", x.batched)


def channel_scale(x: Act, s: Act) -> Act:
    """x * s with s one value per (image, channel): SqueezeExcitation's last line (layers/squeeze.py:60)."""
    x = as_map(x)
    B, H, W, C = x.t.shape
    st = s.t.reshape(B, -1)
    if st.shape[1] != C:
        raise ValueError(f"channel_scale: scale has {st.shape[1]} channels, the map {C}")
    if st.dtype != x.t.dtype:
        s = cast(Act(st, "vec", s.batched), "bf16" if x.t.dtype == torch.bfloat16 else "fp32")
        st = s.t
    if not st.is_contiguous():          # a temporary copy would not be pinned for the recorded launch list
        raise ValueError("channel_scale: the scale must be a contiguous [B, C] tensor")
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_channel_scale_nhwc_fwd", _ptr(x.t), _ptr(st), _ptr(y), B, H * W, C, x.dt, stream_ptr())
    return Act(y, "map", x.batched)


def _batched_keys(key, B: int) -> np.ndarray:
    """uint32 [B, 2]: the per-sample keys of a vmapped call; one key [2] serves an un-batched call (B == 1)."""
    k = np.asarray(key, np.uint32)
    if k.ndim == 1 and k.shape[0] == 2 and B == 1:
        k = k[None]
    if k.shape != (B, 2):
        raise ValueError(f"expected one PRNG key per sample, uint32 [{B}, 2]; got {k.shape}")
    return np.ascontiguousarray(k)


def _keys_dev(key, B: int) -> torch.Tensor:
    """int32 [B, 2] on the device: host keys (numpy, one per sample) are uploaded, keys derived on the device pass through."""
    if isinstance(key, torch.Tensor):
        if tuple(key.shape) != (B, 2) or key.dtype != torch.int32 or not key.is_contiguous():
            raise ValueError(f"expected device keys int32 [{B}, 2], got {key.dtype} {tuple(key.shape)}")
        return key
    return torch.from_numpy(_batched_keys(key, B).view(np.int32)).to(device())


def split_keys(keys, num: int):
    """jax.random.split of batched keys [R, 2] -> [num, R, 2] (child i of every key): numpy in, numpy out (random.split); a
    device tensor in, a device tensor out (mv_prng_split) -- what the per-token keys of a transformer MLP go through."""
    if not isinstance(keys, torch.Tensor):
        from . import random as jr
        return jr.split(np.asarray(keys, np.uint32), num)
    R = keys.shape[0]
    out = empty((num, R, 2), torch.int32)
    _lib.call("mv_prng_split", _ptr(_keys_dev(keys, R)), _ptr(out), R, int(num), 1, stream_ptr())
    return out


def token_keys(key, B: int, N: int) -> torch.Tensor:
    """int32 [B*N, 2] on the device: jax.random.split(key, N) of every sample's key, rows in (sample, token) order -- the keys
    `jax.vmap(self.mlp)(y, key=jrandom.split(keys[2], x.shape[0]))` hands the tokens (vit.py:155)."""
    out = empty((B * N, 2), torch.int32)
    _lib.call("mv_prng_split", _ptr(_keys_dev(key, B)), _ptr(out), B, int(N), 0, stream_ptr())
    return out


def dropout(x: Act, p: float, key, per_row: bool = False) -> Act:
    """eqx.nn.Dropout's training branch: where(bernoulli(key, 1 - p, x.shape), x / (1 - p), 0) per sample, the mask from the
    sample's key in JAX's bit stream (generated on the device, rng.hip).  `per_row`: x is (tokens, features) rows and `key`
    holds one key per ROW (uint32 [B*N, 2], `token_keys`): the reference vmaps the layer over the tokens.  A map is always the
    reference's (C, H, W) array (also behind Swin's Linear2d layers, extensions_2d.py:46-50): the kernel maps NHWC positions to it."""
    if x.kind == "img":
        x = as_map(x)
    B = x.t.shape[0]
    C = x.t.shape[-1]
    if per_row:
        if x.kind != "seq":
            raise ValueError(f"per-row dropout expects (tokens, features) rows, got {x.kind} {tuple(x.t.shape)}")
        B = x.t.numel() // C
    keys = _keys_dev(key, B)
    per = x.t.numel() // B
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_dropout_fwd", _ptr(x.t), _ptr(keys), _ptr(y), B, per, C, 1 if x.kind == "map" else 0,
              float(1.0 - p), x.dt, stream_ptr())
    return Act(y, x.kind, x.batched)


def drop_path(x: Act, p: float, mode: str, key) -> Act:
    """DropPath's training branch (drop_path.py:51-61): noise = bernoulli(key, 1 - p) for the whole sample (mode "global") or one
    draw per entry of the sample's FIRST logical axis (mode "local": channels of a (C,H,W) map, tokens of an (N,D) matrix),
    divided by 1 - p when that is positive; x * noise.  The draws are made on the device from the samples' keys (JAX's bit stream,
    mv_drop_path_noise -> a [B, C] scale); the multiply is the broadcast-scale kernel."""
    if x.kind == "img":
        x = as_map(x)
    B = x.t.shape[0]
    keep = 1.0 - float(p)
    if x.kind == "seq" and mode != "global":
        raise NotImplementedError("DropPath(mode='local') on a (tokens, features) array is not on any model's path")
    C = x.t.shape[-1]
    if keep <= 0.0:                    # p = 1: bernoulli(key, 0) is all False and the reference skips the division
        _keys_dev(key, B)
        s = torch.zeros((B, C), dtype=x.t.dtype, device=device())
    else:
        s = empty((B, C), x.t.dtype)
        _lib.call("mv_drop_path_noise", _ptr(_keys_dev(key, B)), _ptr(s), B, C, 0 if mode == "global" else 1,
                  float(np.float32(keep)), x.dt, stream_ptr())
    xm = x if x.kind == "map" else Act(x.t.reshape(B, -1, 1, C), "map", x.batched)
    y = channel_scale(xm, Act(s, "vec", x.batched))
    return y if x.kind == "map" else Act(y.t.reshape(x.t.shape), x.kind, x.batched)


def patch_merge_ln(x: Act, ln) -> Optional[Act]:
    """norm(_patch_merging_pad(x)) (swin.py:61-65) in one pass over the fp32 map -- the gathered 4C map is never written; None when
    the library has no such path (odd sizes, bf16 stream)."""
    x = as_map(x)
    B, H, W, C = x.t.shape
    dt = compute_dtype()
    if dt != "bf16" or x.t.dtype != torch.float32 or int(np.prod(ln.shape)) != 4 * C or ln.weight is None or ln.bias is None:
        return None
    if not _lib.load().mv_patch_merge_ln_supported(H, W, C, _lib.F32):
        return None
    y = empty((B, H // 2, W // 2, 4 * C), TORCH_DT[dt])
    _lib.call("mv_patch_merge_ln_fwd", _ptr(x.t), _ptr(prep_f32(ln, "weight", ln.weight)), _ptr(prep_f32(ln, "bias", ln.bias)), _ptr(y),
              B, H, W, C, float(ln.eps), _lib.F32, DT[dt], stream_ptr())
    return Act(y, "map", x.batched)


def patch_merge_gather(x: Act) -> Act:
    x = as_map(x)
    B, H, W, C = x.t.shape
    y = empty((B, (H + 1) // 2, (W + 1) // 2, 4 * C), x.t.dtype)
    _lib.call("mv_patch_merge_gather_nhwc", _ptr(x.t), _ptr(y), B, H, W, C, x.dt, stream_ptr())
    return Act(y, "map", x.batched)


# ------------------------------------------------------------------ element-wise (unfused call sites)
def _canon(x: Act) -> Act:
    return as_map(x) if x.kind == "img" else (x if x.kind == "map" else as_rows(x))


def eltwise(x: Act, act: str) -> Act:
    x = _canon(x)
    y = empty(tuple(x.t.shape), x.t.dtype)
    _lib.call("mv_eltwise_fwd", _ptr(x.t), _ptr(y), x.t.numel(), ACT[act], x.dt, stream_ptr())
    return Act(y, x.kind, x.batched)


def add(a: Act, b: Act, act=None) -> Act:
    a, b = _canon(a), _canon(b)
    if a.t.dtype != b.t.dtype and {a.t.dtype, b.t.dtype} == {torch.float32, torch.bfloat16}:
        # an fp32 residual stream + a compute-dtype branch (the un-fused x + drop_path(f(x)) of training mode): add in fp32
        a, b = (a if a.t.dtype == torch.float32 else cast(a, "fp32")), (b if b.t.dtype == torch.float32 else cast(b, "fp32"))
    if tuple(a.t.shape) != tuple(b.t.shape) or a.t.dtype != b.t.dtype:
        raise ValueError(f"add: mismatched operands {a} vs {b}")
    y = empty(tuple(a.t.shape), a.t.dtype)
    _lib.call("mv_add_fwd", _ptr(a.t), _ptr(b.t), _ptr(y), a.t.numel(), ACT[act], a.dt, stream_ptr())
    return Act(y, a.kind, a.batched)


def first_row(x: Act) -> Act:
    """x[0] of a seq [B,N,D] -> vec [B,D] (a strided gather done by the cast kernel per image)."""
    B, N, D = x.t.shape
    y = empty((B, D), x.t.dtype)
    for b in range(B):
        _lib.call("mv_cast", x.t[b].data_ptr(), y[b].data_ptr(), D, x.dt, x.dt, stream_ptr())
    return Act(y, "vec", x.batched)


# ------------------------------------------------------------------ gradients (eqxvision_amd/grad.py)
# Under `filter_value_and_grad` the entry points below run their differentiable fp32 twins; everywhere else they are untouched.
from . import grad as _grad  # noqa: E402

for _name in _grad.HOOKED:
    globals()[_name] = _grad.hook(_name, globals()[_name])
del _name
