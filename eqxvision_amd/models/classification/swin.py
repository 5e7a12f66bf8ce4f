"""Swin Transformer V1 (reference models/classification/swin.py:23-66, 90-366, 526-578, 640-830).

Same fields / constructors.  V2 (cosine attention + cpb_mlp) is out of scope.
Device lowering of one block (reference :572-578):
  LayerNorm2d rows | qkv GEMM(+bias) | window attention (roll / partition / bias / mask / softmax / PV /
  reverse folded into addressing) | proj GEMM(+bias,+residual) | LayerNorm2d | fc1 GEMM(+gelu) |
  fc2 GEMM(+residual)
Feature maps stay NHWC, which is exactly the (H,W,C) view `_shifted_window_attention` transposes to
(reference :106), so no transposes exist on the device.
"""
from __future__ import annotations

import warnings
from functools import partial
from typing import Any, Callable, List, Optional

import numpy as np

from ... import nn, ops
from ... import random as jr
from ..._act import Act, head_fp32, residual_fp32
from ..._module import Module
from ...layers import DropPath, LayerNorm2d, Linear2d, MlpProjection
from ...nn import boundary
from ...utils import load_torch_weights


def _get_relative_position_bias(table: np.ndarray, index: np.ndarray, window_size: List[int]) -> np.ndarray:
    """table[index] -> (heads, N, N)  (reference :34-43); negative indices wrap like jnp indexing."""
    n = window_size[0] * window_size[1]
    idx = np.asarray(index).reshape(-1).astype(np.int64)
    b = np.asarray(table, np.float32)[idx]
    return np.ascontiguousarray(b.reshape(n, n, -1).transpose(2, 0, 1))


class _PatchMerging(Module):
    reduction: Linear2d
    norm: Callable

    def __init__(self, dim: int, norm_layer: Callable = LayerNorm2d, *, key=None):
        self.norm = norm_layer(4 * dim)
        self.reduction = Linear2d(4 * dim, 2 * dim, use_bias=False, key=key)

    @boundary
    def __call__(self, x, *, key=None):                                # reference :61-65
        y = ops.patch_merge_ln(x, self.norm) if isinstance(self.norm, nn.LayerNorm) else None     # gather + LayerNorm, one pass
        if y is None:
            y = self.norm(ops.patch_merge_gather(x))                   # _patch_merging_pad (:23-31), then the norm
        x = y
        if type(self.reduction) is Linear2d:      # produces the residual stream: split-precision weights (ops.linear_split)
            return ops.linear_split(ops.as_map(x), self.reduction, out_fp32=residual_fp32())
        return self.reduction(x)


class _ShiftedWindowAttention(Module):
    window_size: List[int]
    shift_size: List[int]
    num_heads: int
    attention_dropout: float
    dropout: float
    relative_position_bias_table: np.ndarray
    relative_position_index: np.ndarray
    qkv: nn.Linear
    proj: nn.Linear

    def __init__(self, dim: int, window_size: List[int], shift_size: List[int], num_heads: int, qkv_bias: bool = True,
                 proj_bias: bool = True, attention_dropout: float = 0.0, dropout: float = 0.0, *, key=None):
        if len(window_size) != 2 or len(shift_size) != 2:
            raise ValueError("window_size and shift_size must be of length 2")
        keys = jr.split(key if key is not None else jr.PRNGKey(0), 3)
        self.window_size = list(window_size)
        self.shift_size = list(shift_size)
        self.num_heads = num_heads
        self.attention_dropout = attention_dropout
        self.dropout = dropout
        self.qkv = Linear2d(dim, dim * 3, use_bias=qkv_bias, key=keys[0])
        self.proj = Linear2d(dim, dim, use_bias=proj_bias, key=keys[1])
        self.relative_position_bias_table = self.define_relative_position_bias_table(key=keys[2])
        self.relative_position_index = self.define_relative_position_index()

    def define_relative_position_bias_table(self, key):
        # reference :303-312: truncated_normal(lower=2, upper=2) -> the constant 2.0 (softmax-invariant)
        n = (2 * self.window_size[0] - 1) * (2 * self.window_size[1] - 1)
        return jr.truncated_normal(key, 2, 2, (n, self.num_heads))

    def define_relative_position_index(self):
        # reference :314-335 quirk: the torchvision-style index is built and DISCARDED;
        # what is returned is relative_coords.sum(-1) in [-(Wh+Ww-2), Wh+Ww-2].  Reproduced as data; a
        # pretrained checkpoint overwrites it with torchvision's index (Appendix C-1).
        ch, cw = np.arange(self.window_size[0]), np.arange(self.window_size[1])
        coords = np.stack(np.meshgrid(ch, cw, indexing="ij")).reshape(2, -1)
        rel = coords[:, :, None] - coords[:, None, :]
        return rel.transpose(1, 2, 0).sum(-1).reshape(-1).astype(np.int32)

    def get_relative_position_bias(self) -> np.ndarray:
        return _get_relative_position_bias(self.relative_position_bias_table, self.relative_position_index,
                                           self.window_size)

    def _bias_dev(self):
        return ops.swin_rel_bias(self)

    def _live(self) -> bool:
        """The reference's `_func_dropout` (swin.py:17-20, 227, 233) has no inference switch: a non-zero rate drops in EVERY mode."""
        return self.attention_dropout > 0 or self.dropout > 0

    def _forward(self, x: Act, residual: Optional[Act] = None, norm=None, key=None, precise: Optional[bool] = None) -> Act:
        """`norm` given: x is the UN-normalised input and the LayerNorm is folded into the qkv Linear where possible."""
        x = ops.as_map(x)
        B, Hf, Wf, C = x.t.shape
        if Hf % self.window_size[0] or Wf % self.window_size[1]:
            raise ValueError(f"feature map {Hf}x{Wf} is not a multiple of the window {self.window_size} "
                             "(the reference does not pad either, swin.py:782-790)")
        if precise is None:
            precise = ops.swin_precise(C)        # widths off the fused kernels (swin_b): split-precision block Linears
        if precise:
            qkv = ops.linear_split(x if norm is None else ops.layernorm(x, norm), self.qkv)
        else:
            qkv = ops.linear(x, self.qkv) if norm is None else ops.ln_linear(x, norm, self.qkv)      # reference :151-153
        if not self._live():
            a = ops.swin_window_attention(qkv, self._bias_dev(), self.num_heads, self.window_size, self.shift_size)
            if precise:
                return ops.linear_split(a, self.proj, residual=residual)
            return ops.linear(a, self.proj, residual=residual)         # reference :232 (+ the block's residual)
        if key is None:
            raise RuntimeError("Swin attention_dropout / dropout > 0 requires a key (drawn in every mode, swin.py:17-20)")
        kd = ops._keys_dev(key, B)                                     # ONE key for both draws (reference :227, :233)
        drop = (self.attention_dropout, kd) if self.attention_dropout > 0 else None
        a = ops.swin_window_attention(qkv, self._bias_dev(), self.num_heads, self.window_size, self.shift_size, drop=drop)
        y = ops.linear(a, self.proj)
        if self.dropout > 0:
            y = ops.dropout_windows(y, self.dropout, kd, self.window_size, self.shift_size)
        return y if residual is None else ops.add(residual, y)

    @boundary
    def __call__(self, x, *, key=None):
        return self._forward(x, key=key)


class _SwinTransformerBlock(Module):
    norm1: Callable
    attn: Module
    stochastic_depth: DropPath
    norm2: Callable
    mlp: MlpProjection

    def __init__(self, dim: int, num_heads: int, window_size: List[int], shift_size: List[int], mlp_ratio: float = 4.0,
                 dropout: float = 0.0, attention_dropout: float = 0.0, stochastic_depth_prob: float = 0.0,
                 norm_layer: Callable[..., Module] = LayerNorm2d,
                 attn_layer: Callable[..., Module] = _ShiftedWindowAttention, *, key=None):
        keys = jr.split(key if key is not None else jr.PRNGKey(0), 2)
        self.norm1 = norm_layer(dim)
        self.attn = attn_layer(dim, window_size, shift_size, num_heads, attention_dropout=attention_dropout,
                               dropout=dropout, key=keys[0])
        self.stochastic_depth = DropPath(stochastic_depth_prob, mode="local")
        self.norm2 = norm_layer(dim)
        self.mlp = MlpProjection(dim, int(dim * mlp_ratio), dim, lin_layer=Linear2d, act_layer=nn.gelu, drop=dropout,
                                 key=keys[1])

    @boundary
    def __call__(self, x, *, key=None):                                # reference :572-578
        x = ops.as_map(x)
        sd = self.stochastic_depth
        attn_live = bool(getattr(self.attn, "_live", lambda: False)())
        mlp_live = isinstance(self.mlp, MlpProjection) and self.mlp._live()
        if (sd.inference or sd.p == 0.0) and not attn_live and not mlp_live:
            # `_deep_stage` (set by SwinTransformer on the blocks of a stage deeper than 6: swin_s / swin_b's 18-block stage 2): the bf16
            # rounding of the block Linears' weights adds up over the blocks -- swin_s sat at 9.97e-3 of the 1e-2 bound on the fused
            # kernels -- so those blocks take the split-precision path (ops.swin_precise) like the widths without fused kernels
            precise = ops.swin_precise(x.t.shape[-1]) or (getattr(self, "_deep_stage", False) and ops.swin_precise(0))
            if type(self.attn) is _ShiftedWindowAttention and isinstance(self.norm1, nn.LayerNorm):
                y = None if precise else ops.swin_block_attention(x, self.norm1, self.attn)     # the whole attention half, one workgroup per window
                x = y if y is not None else self.attn._forward(x, residual=x, norm=self.norm1, precise=precise)
            else:
                x = self.attn._forward(self.norm1(x), residual=x)
            if isinstance(self.mlp, MlpProjection):
                y = None if precise else ops.ln_mlp(x, self.norm2, self.mlp)   # one launch where the weights fit in LDS (stage 0)
                if y is not None:
                    return y
                if isinstance(self.norm2, nn.LayerNorm):
                    return self.mlp._forward(x, residual=x, norm=self.norm2, precise=precise)
            return self.mlp._forward(self.norm2(x), residual=x)
        if key is None:
            raise RuntimeError("stochastic depth outside inference mode / Swin dropout > 0 requires a key")
        keys = jr.split(ops._batched_keys(key, x.t.shape[0]), 4)       # reference :573: attention, path, MLP, path
        x = ops.add(x, sd(self.attn._forward(self.norm1(x), key=keys[0]), key=keys[1]))
        m = self.mlp._forward(self.norm2(x), keys=keys[2]) if mlp_live else self.mlp._forward(self.norm2(x))
        return ops.add(x, sd(m, key=keys[3]))


class SwinTransformer(Module):
    features: nn.Sequential
    norm: Callable
    avgpool: nn.AdaptiveAvgPool2d
    head: nn.Linear

    def __init__(self, patch_size: List[int], embed_dim: int, depths: List[int], num_heads: List[int],
                 window_size: List[int], mlp_ratio: float = 4.0, dropout: float = 0.0, attention_dropout: float = 0.0,
                 stochastic_depth_prob: float = 0.1, num_classes: int = 1000, norm_layer: Callable = None,
                 block: Module = None, downsample_layer: Module = None, *, key=None):
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, 2)
        if block is None:
            block = _SwinTransformerBlock
        if norm_layer is None:
            norm_layer = partial(LayerNorm2d, eps=1e-5)
        if downsample_layer is None:
            downsample_layer = _PatchMerging
        stack: List[Module] = [nn.Sequential([
            nn.Conv2d(3, embed_dim, kernel_size=(patch_size[0], patch_size[1]),
                      stride=(patch_size[0], patch_size[1]), key=keys[0]),
            norm_layer(embed_dim),
        ])]
        total = sum(depths)
        bid = 0
        for i_stage in range(len(depths)):
            stage: List[Module] = []
            dim = embed_dim * 2 ** i_stage
            for i_layer in range(depths[i_stage]):
                keys = jr.split(keys[1], 2)
                sd_prob = stochastic_depth_prob * float(bid) / (total - 1)
                stage.append(block(dim, num_heads[i_stage], window_size=window_size,
                                   shift_size=[0 if i_layer % 2 == 0 else w // 2 for w in window_size],
                                   mlp_ratio=mlp_ratio, dropout=dropout, attention_dropout=attention_dropout,
                                   stochastic_depth_prob=sd_prob, norm_layer=norm_layer, key=keys[0]))
                bid += 1
            if depths[i_stage] > 6:                # a non-field attribute: travels with tree copies, invisible to the leaf order
                for blk in stage:
                    object.__setattr__(blk, "_deep_stage", True)
            stack.append(nn.Sequential(stage))
            if i_stage < len(depths) - 1:
                keys = jr.split(keys[1], 2)
                stack.append(downsample_layer(dim, norm_layer, key=keys[0]))
        self.features = nn.Sequential(stack)
        num_features = embed_dim * 2 ** (len(depths) - 1)
        self.norm = norm_layer(num_features)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.head = nn.Linear(num_features, num_classes, key=keys[1])

    def _features(self, x, key=None):
        """self.features(x); the patch-embed LayerNorm2d starts the fp32 residual stream when enabled.  `key`: split per layer
        like nn.Sequential does (stochastic depth in training mode)."""
        L = self.features.layers
        ks = [None] * len(L) if key is None else list(jr.split(key, len(L)))
        first = L[0]
        if (residual_fp32() and isinstance(first, nn.Sequential) and len(first) == 2
                and type(first.layers[0]) is nn.Conv2d and isinstance(first.layers[1], nn.LayerNorm)):
            y = ops.patch4_ln(x, first.layers[0], first.layers[1])  # patch embedding + its norm, one launch, fp32 throughout
            if y is not None:
                x = y
            else:
                y = ops.conv2d_entry_split(x, first.layers[0])      # patch embedding: split-precision weights
                x = y if y is not None else ops.conv2d(x, first.layers[0])
                x = ops.layernorm(x, first.layers[1], out_fp32=True)
            for layer, k in zip(L[1:], ks[1:]):
                x = layer(x, key=k)
            return x
        return self.features(x, key=key)

    @boundary
    def __call__(self, x, *, key=None):                                # reference :760-772
        x = self._features(x, None if key is None else jr.split(key, 2)[0])
        if head_fp32() and isinstance(self.norm, nn.LayerNorm) and type(self.avgpool) is nn.AdaptiveAvgPool2d:
            x = ops.layernorm(x, self.norm, out_fp32=True)             # reference :768-771 with fp32 features
            x = ops.adaptive_avgpool2d(x, self.avgpool.target_shape, out_fp32=True)
        else:
            x = self.norm(x)
            x = self.avgpool(x)
        x = ops.flatten(x)
        return ops.linear_head(x, self.head)


def _swin_transformer(arch, patch_size, embed_dim, depths, num_heads, window_size, stochastic_depth_prob,
                      torch_weights, **kwargs: Any) -> SwinTransformer:
    warnings.warn("Currently, dynamic padding of the input is not supported! "
                  "Please make sure that the input is a multiple of window_size.")
    model = SwinTransformer(patch_size=patch_size, embed_dim=embed_dim, depths=depths, num_heads=num_heads,
                            window_size=window_size, stochastic_depth_prob=stochastic_depth_prob, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


def swin_t(torch_weights: str = None, **kwargs: Any) -> SwinTransformer:
    return _swin_transformer("swin_t", [4, 4], 96, [2, 2, 6, 2], [3, 6, 12, 24], [7, 7], 0.2, torch_weights, **kwargs)


def swin_s(torch_weights: str = None, **kwargs: Any) -> SwinTransformer:
    return _swin_transformer("swin_s", [4, 4], 96, [2, 2, 18, 2], [3, 6, 12, 24], [7, 7], 0.3, torch_weights, **kwargs)


def swin_b(torch_weights: str = None, **kwargs: Any) -> SwinTransformer:
    return _swin_transformer("swin_b", [4, 4], 128, [2, 2, 18, 2], [4, 8, 16, 32], [7, 7], 0.5, torch_weights, **kwargs)
