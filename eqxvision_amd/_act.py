"""Device activation handle and the precision / allocation context.

The reference's modules are single-sample functions `(C,H,W) -> ...` batched by the caller with
`jax.vmap` (reference README.md:37-40).  Here the batch axis is physical from the start: an `Act`
carries a torch device tensor whose leading axis is the batch (1 when the caller did not vmap)
plus the *logical single-sample* view the reference code would see.

Physical layouts in HBM:
  kind "img": raw user images, NCHW [B,C,H,W], fp32 or bf16 (only at the network entry)
  kind "map": feature maps, NHWC [B,H,W,C]  -- channel-contiguous so that the K axis of every
              implicit GEMM is contiguous for 16-byte MFMA fragments
  kind "seq": token / row matrices [B,N,D]
  kind "vec": feature vectors [B,D]
torch is used for device memory and streams only; every computation goes through the C ABI.
"""
from __future__ import annotations

import contextlib
from typing import Tuple

import numpy as np
import torch

from . import _lib

import threading

_state = {"dtype": "bf16", "residual_fp32": True, "head_fp32": True, "split_weights": True}   # process-wide numerics configuration
_tls = threading.local()                                  # .keep: the allocation pin list of THIS thread's trace

DT = {"bf16": _lib.BF16, "fp32": _lib.F32}
TORCH_DT = {"bf16": torch.bfloat16, "fp32": torch.float32}


def compute_dtype() -> str:
    return _state["dtype"]


def set_compute_dtype(name: str):
    """'bf16' (bf16 storage, fp32 accumulate/epilogue -- the MI355X headline path) or 'fp32'."""
    if name not in DT:
        raise ValueError(f"compute dtype must be 'bf16' or 'fp32', got {name!r}")
    _state["dtype"] = name


def residual_fp32() -> bool:
    """Transformer residual streams (ViT tokens, Swin maps) are kept in fp32 between blocks while every
    GEMM / attention operand stays bf16: the GEMM epilogues add the fp32 residual and store fp32, the
    LayerNorms read fp32 and emit bf16.  Costs ~2x bytes on the (compute-bound) residual tensors and buys
    ~2x lower logit error over 12 blocks.  Irrelevant in fp32 mode."""
    return _state["residual_fp32"] and _state["dtype"] == "bf16"


def set_residual_fp32(on: bool):
    _state["residual_fp32"] = bool(on)


def head_fp32() -> bool:
    """Classifier heads (pooled / cls features -> logits: resnet.py:354-356, vit.py:272-273, swin.py:768-771) run in fp32 on
    the exact-fp32 MFMA while the rest of the network stays bf16: the logits are the tested quantity, and rounding the pooled
    features + the head weights to bf16 alone costs 3-5e-3 of the 1e-2 budget.  ~0.3 GFLOP: free."""
    return _state["head_fp32"] and _state["dtype"] == "bf16"


def set_head_fp32(on: bool):
    _state["head_fp32"] = bool(on)


def split_weights() -> bool:
    """Layers that PRODUCE the residual stream (Swin patch embedding and patch merging, swin.py:705-711 / 61-65) carry their
    fp32 weights as two bf16 terms (hi + lo) and run two products: their weight rounding is not damped by a residual add and
    dominates the bf16 logit error of swin_t (1.4e-2 -> 8e-3 with this and the fp32 head)."""
    return _state["split_weights"] and _state["dtype"] == "bf16"


def set_split_weights(on: bool):
    _state["split_weights"] = bool(on)


@contextlib.contextmanager
def precision(name: str):
    old = _state["dtype"]
    set_compute_dtype(name)
    try:
        yield
    finally:
        _state["dtype"] = old


def device() -> torch.device:
    if not torch.cuda.is_available():
        raise _lib.MVError("eqxvision_amd needs an MI355X (no HIP device visible); there is no CPU fallback")
    return torch.device("cuda", torch.cuda.current_device())


def stream_ptr() -> int:
    return torch.cuda.current_stream().cuda_stream


def empty(shape, dtype: torch.dtype) -> torch.Tensor:
    t = torch.empty(tuple(shape), dtype=dtype, device=device())
    k = getattr(_tls, "keep", None)
    if k is not None:          # a recording owns every intermediate so replayed pointers stay valid
        k.append(t)
    return t


@contextlib.contextmanager
def keep_alive(lst):
    old = getattr(_tls, "keep", None)
    _tls.keep = lst
    try:
        yield
    finally:
        _tls.keep = old


@contextlib.contextmanager
def collect_replay_hooks(lst):
    """While a forward is being recorded: callables that must run after EVERY replay of the recording (host-side bookkeeping of
    device-side effects, e.g. "the BatchNorm statistics on the device changed")."""
    old = getattr(_tls, "hooks", None)
    _tls.hooks = lst
    try:
        yield
    finally:
        _tls.hooks = old


def on_replay(fn) -> None:
    h = getattr(_tls, "hooks", None)
    if h is not None:
        h.append(fn)


class Act:
    __slots__ = ("t", "kind", "batched", "pre", "node", "sub", "ln")

    def __init__(self, t: torch.Tensor, kind: str, batched: bool):
        self.t = t
        self.kind = kind
        self.batched = batched
        self.pre = None     # (module, Act): the result of applying `module` (+ its norm + relu) to this activation was
                            # already produced by the launch that produced it (ops.conv1x1_chain)
        self.node = None    # under filter_value_and_grad: the autograd node that produced this activation (grad.py)
        self.ln = None      # (low plane, per-row statistics pieces): `t` is the HIGH bf16 plane of a residual stream kept as two planes
                            # (hi = bf16(y), lo = bf16(y - hi)) -- what a LayerNorm + Linear pair behind these rows needs instead of
                            # a LayerNorm launch (ops.linear_lnout -> ops.linear_lnin; ops.stream_f32 gives fp32 rows back)
        self.sub = None     # s: `t` holds only the pixels (s i, s j) of the logical map -- written that way because the one consumer
                            # left is a stride-s pointwise convolution (ops.conv1x1_chain(..., sub=), ops.conv1x1_dual)

    @property
    def B(self) -> int:
        return self.t.shape[0]

    @property
    def shape(self) -> Tuple[int, ...]:
        """Logical single-sample shape, as the reference module code would see it."""
        s = tuple(self.t.shape[1:])
        if self.kind == "map":      # physical (H,W,C) -> logical (C,H,W)
            return (s[2], s[0], s[1])
        return s

    @property
    def dt(self) -> int:
        return _lib.BF16 if self.t.dtype == torch.bfloat16 else _lib.F32

    def __repr__(self):
        return f"Act({self.kind}, logical={self.shape}, B={self.B}, {self.t.dtype}, batched={self.batched})"


def _to_device_f32(x) -> torch.Tensor:
    if isinstance(x, torch.Tensor):
        t = x
    else:
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(x)))
    if t.dtype not in (torch.float32, torch.bfloat16):
        t = t.to(torch.float32)
    return t.to(device(), non_blocking=True).contiguous()


def wrap(x, batched: bool) -> Act:
    """User array -> Act.  Rank (without the batch axis) picks the kind: 3 = image (C,H,W),
    2 = rows (N,D), 1 = vector (D,)."""
    if isinstance(x, Act):
        return x
    t = _to_device_f32(x)
    if not batched:
        t = t.unsqueeze(0)
    r = t.dim() - 1
    if r == 3:
        return Act(t, "img", batched)
    if r == 2:
        return Act(t, "seq", batched)
    if r == 1:
        return Act(t, "vec", batched)
    raise ValueError(f"unsupported input rank {r} (shape {tuple(t.shape)})")


def is_act(x) -> bool:
    return isinstance(x, Act)
