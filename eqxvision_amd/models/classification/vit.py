"""Vision Transformer (reference models/classification/vit.py:15-404; DINO port).

Same classes / fields / constructors / `__call__` signatures.  Device lowering of one block
(`_VitBlock.__call__`, reference :139-157) = 7 launches:
  LayerNorm | qkv GEMM(+bias) | fused attention (QK^T, softmax, PV) | proj GEMM(+bias,+residual)
  LayerNorm | fc1 GEMM(+bias,+gelu) | fc2 GEMM(+bias,+residual)
and 5 in inference where the rows fill the 256 x 256 GEMM tiles (round 6): the two LayerNorms become per-row factors in the
epilogues of the GEMMs on either side of them (`_VitBlock._ln_fold`, ops.linear_lnout / linear_lnin).
"""
from __future__ import annotations

from typing import Optional, Sequence, Tuple, Union

import numpy as np
import torch

from ... import nn, ops
from ... import random as jr
from ..._act import Act, head_fp32, residual_fp32
from ..._module import Module
from ...layers import DropPath, MlpProjection, PatchEmbed
from ...nn import boundary
from ...utils import load_torch_weights


class _VitAttention(Module):
    num_heads: int
    scale: float
    qkv: nn.Linear
    attn_drop: nn.Dropout
    proj: nn.Linear
    proj_drop: nn.Dropout

    def __init__(self, dim: int, num_heads: int = 8, qkv_bias: bool = False, qk_scale=None, attn_drop: float = 0.0,
                 proj_drop: float = 0.0, *, key=None):
        keys = jr.split(key if key is not None else jr.PRNGKey(0), 2)
        self.num_heads = num_heads
        head_dim = dim // num_heads
        self.scale = qk_scale or head_dim ** -0.5
        self.qkv = nn.Linear(dim, dim * 3, use_bias=qkv_bias, key=keys[0])
        self.attn_drop = nn.Dropout(attn_drop)
        self.proj = nn.Linear(dim, dim, key=keys[1])
        self.proj_drop = nn.Dropout(proj_drop)

    def _live(self) -> bool:
        return nn.dropout_live(self.attn_drop) or nn.dropout_live(self.proj_drop)

    def _forward(self, x: Act, residual: Optional[Act] = None, need_probs: bool = True, key=None):
        x = ops.as_rows(x)          # GEMM operand: compute dtype
        if x.kind != "seq":
            raise ValueError(f"_VitAttention expects (tokens, dim), got {x.shape}")
        drop, pkey = None, None
        if self._live():            # training mode: keys = split(key, 2) (reference :63); [0] -> attn_drop (:71), [1] -> proj_drop (:75)
            if key is None:
                raise RuntimeError("Dropout requires a key when running in non-deterministic mode.")
            ks = jr.split(ops._batched_keys(key, x.t.shape[0]), 2)
            if nn.dropout_live(self.attn_drop):
                drop = (self.attn_drop.p, ks[0])
            if nn.dropout_live(self.proj_drop):
                pkey = ks[1]
        y, probs = ops.qkv_attention(x, self.qkv, self.num_heads, self.scale, need_probs, drop=drop)   # reference :64-73
        if pkey is None:
            y = ops.linear(y, self.proj, residual=residual)            # reference :74 (+ the block's residual)
        else:
            y = self.proj_drop(ops.linear(y, self.proj), key=pkey)     # reference :74-75
            if residual is not None:
                y = ops.add(residual, y)
        attn = None
        if probs is not None:                                          # reference returns (1, heads, N, N) per sample
            B, H, N, _ = probs.shape
            attn = Act(probs.reshape(B, 1, H, N, N), "raw", x.batched)
        return y, attn

    @boundary
    def __call__(self, x, *, key=None):
        return self._forward(x, key=key)


class _VitBlock(Module):
    norm1: Module
    attn: _VitAttention
    drop_path: DropPath
    norm2: Module
    mlp: MlpProjection

    def __init__(self, dim, num_heads, mlp_ratio=4.0, qkv_bias=False, qk_scale=None, drop=0.0, attn_drop=0.0,
                 drop_path=0.0, act_layer=nn.gelu, norm_layer=nn.LayerNorm, *, key):
        keys = jr.split(key, 2)
        self.norm1 = norm_layer(dim)
        self.attn = _VitAttention(dim, num_heads=num_heads, qkv_bias=qkv_bias, qk_scale=qk_scale,
                                  attn_drop=attn_drop, proj_drop=drop, key=keys[0])
        self.drop_path = DropPath(float(drop_path)) if drop_path > 0.0 else nn.Identity()
        self.norm2 = norm_layer(dim)
        self.mlp = MlpProjection(in_features=dim, hidden_features=int(dim * mlp_ratio), act_layer=act_layer,
                                 drop=drop, key=keys[1])

    def _deterministic(self) -> bool:
        dp = self.drop_path
        return ((isinstance(dp, nn.Identity) or dp.inference or dp.p == 0.0)
                and not self.attn._live() and not self.mlp._live())

    def _ln_fold(self, x: Act):
        """The inference form without LayerNorm launches, where the library has it for this many rows: proj and fc2 keep the
        residual stream as two bf16 planes (+ row statistics), fc1 and qkv take the LayerNorm in their epilogue (ops.linear_lnout /
        linear_lnin).  -> (fold norm2 -> fc1, fold norm1 -> qkv) for rows x."""
        split = ops.is_split_stream(x)
        if x.kind != "seq" or x.t.dim() != 3 or not (split or (x.t.dtype == torch.float32 and x.ln is None)):
            return False, False
        mlp, at = self.mlp, self.attn
        if not (type(mlp) is MlpProjection and type(mlp.fc1) is nn.Linear and type(mlp.fc2) is nn.Linear
                and nn.act_name(mlp.act) in ("gelu", "relu") and type(at.qkv) is nn.Linear and type(at.proj) is nn.Linear):
            return False, False
        B, N, D = x.t.shape
        M = B * N
        if not (at.proj.out_features == D and mlp.fc2.out_features == D and mlp.fc1.in_features == D and D % at.num_heads == 0):
            return False, False
        inner = (ops.linear_lnout_available(M, at.proj) and ops.linear_lnout_available(M, mlp.fc2)
                 and ops.linear_lnin_available(M, self.norm2, mlp.fc1))
        first = inner and split and ops.linear_lnin_available(M, self.norm1, at.qkv, N, D // at.num_heads)
        return inner, first

    @boundary
    def __call__(self, x, return_attention=False, *, key=None):        # reference :139-157
        return self._forward(x, return_attention, key=key)

    def _forward(self, x, return_attention=False, key=None, feeds_block=False):
        """`feeds_block`: the rows returned go into another _VitBlock (VisionTransformer.__call__) -- they may then stay a split
        stream (what its norm1 -> qkv pair takes)."""
        x = ops.as_rows(x, keep_fp32=True)      # the residual stream may be fp32 (see _act.residual_fp32)
        if self._deterministic() and not return_attention:
            inner, first = self._ln_fold(x)
            if inner:
                at = self.attn
                if first:
                    a = ops.qkv_attention_ln(x, self.norm1, at.qkv, at.num_heads, at.scale)
                else:
                    a, _ = ops.qkv_attention(ops.as_rows(self.norm1(ops.stream_f32(x))), at.qkv, at.num_heads, at.scale, False)
                x = ops.linear_lnout(a, at.proj, residual=x)                               # reference :74 + :150
                h = ops.linear_lnin(x, self.norm2, self.mlp.fc1, act=nn.act_name(self.mlp.act))
                return ops.linear_lnout(h, self.mlp.fc2, residual=x, out_split=feeds_block)
        x = ops.stream_f32(x)
        y = self.norm1(x)
        if self._deterministic():          # x + Identity(y): fold the adds into the GEMM epilogues
            if return_attention:
                _, attn = self.attn._forward(y, need_probs=True)
                return attn
            x, _ = self.attn._forward(y, residual=x, need_probs=False)
            return self.mlp._forward(self.norm2(x), residual=x)
        # training mode with a live Dropout / DropPath: keys = split(key, 4) (reference :148) -> attention, drop_path, the
        # tokens' MLP keys (split again per token, :155), drop_path
        B, N = x.t.shape[0], x.t.shape[1]
        keys = [None] * 4 if key is None else jr.split(ops._batched_keys(key, B), 4)
        y, attn = self.attn._forward(y, need_probs=bool(return_attention), key=keys[0])
        if return_attention:
            return attn
        x = ops.add(x, self.drop_path(y, key=keys[1]))
        mkeys = ops.token_keys(keys[2], B, N) if self.mlp._live() and keys[2] is not None else None
        y = self.mlp._forward(self.norm2(x), keys=mkeys, per_row=True)
        return ops.add(x, self.drop_path(y, key=keys[3]))


class VisionTransformer(Module):
    num_features: int
    cls_token: np.ndarray
    pos_embed: np.ndarray
    patch_embed: PatchEmbed
    pos_drop: nn.Dropout
    blocks: Sequence[_VitBlock]
    norm: Module
    fc: nn.Linear
    inference: bool

    def __init__(self, img_size: Union[int, Tuple[int]] = 224, patch_size: Union[int, Tuple[int]] = 16,
                 in_chans: int = 3, num_classes: int = 0, embed_dim: int = 768, depth: int = 12, num_heads: int = 12,
                 mlp_ratio: float = 4.0, qkv_bias: bool = True, qk_scale=None, drop_rate=0.0, attn_drop_rate=0.0,
                 drop_path_rate=0.0, norm_layer=nn.LayerNorm, *, key=None):
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, depth + 3)
        self.inference = False
        self.num_features = embed_dim
        self.patch_embed = PatchEmbed(img_size=img_size, patch_size=patch_size, in_chans=in_chans, embed_dim=embed_dim)
        num_patches = self.patch_embed.num_patches
        # unit-scale truncated normals, no 0.02 factor (reference :229-234)
        self.cls_token = jr.truncated_normal(keys[0], -2, 2, (1, embed_dim))
        self.pos_embed = jr.truncated_normal(keys[1], -2, 2, (num_patches + 1, embed_dim))
        self.pos_drop = nn.Dropout(p=drop_rate)
        dpr = [float(v) for v in np.linspace(0, drop_path_rate, depth)]
        self.blocks = [
            _VitBlock(dim=embed_dim, num_heads=num_heads, mlp_ratio=mlp_ratio, qkv_bias=qkv_bias, qk_scale=qk_scale,
                      drop=drop_rate, attn_drop=attn_drop_rate, drop_path=dpr[i], norm_layer=norm_layer,
                      key=keys[i + 1])
            for i in range(depth)
        ]
        self.norm = norm_layer(embed_dim)
        self.fc = nn.Identity() if num_classes == 0 else nn.Linear(embed_dim, num_classes, key=keys[-1])

    def _tokens(self, x: Act) -> Act:
        """patch_embed -> concat(cls, x) + pos_embed (reference :268-269) in one implicit-GEMM launch
        (+ a B x D kernel for the cls row)."""
        pe = self.patch_embed
        if x.kind == "img" and pe.flatten and isinstance(pe.norm, nn.Identity):
            pe._check(x)
            cls = ops.prep_f32(self, "cls_token", self.cls_token.reshape(-1))
            pos = ops.prep_f32(self, "pos_embed", self.pos_embed)
            t = ops.patch_embed_tokens(x, pe.proj, cls, pos, 1, out_fp32=residual_fp32())
            return ops.cast(t, "fp32") if residual_fp32() else t         # no-op when the rows were written in fp32
        raise NotImplementedError("VisionTransformer expects a raw (C,H,W) image and the default PatchEmbed")

    def _head(self, x: Act) -> Act:
        # reference :272-273 normalises every token and keeps x[0]; only the cls row is needed
        if type(self.norm) is nn.LayerNorm:
            cls = ops.layernorm_first_row(x, self.norm, out_fp32=head_fp32() and not isinstance(self.fc, nn.Identity))
        else:
            x = self.norm(x)
            B, N, D = x.t.shape
            cls = ops.first_row(x)
        if isinstance(self.fc, nn.Identity):
            return cls
        return ops.linear_head(cls, self.fc)

    @boundary
    def __call__(self, x, *, key=None):                                # reference :261-273
        x = self._tokens(x)
        keys = [None] * len(self.blocks) if key is None else jr.split(key, len(self.blocks))     # reference :267
        for i, (blk, k) in enumerate(zip(self.blocks, keys)):
            if type(blk) is _VitBlock:
                x = blk._forward(x, key=k, feeds_block=i + 1 < len(self.blocks) and type(self.blocks[i + 1]) is _VitBlock)
            else:
                x = blk(x, key=k)
        return self._head(x)

    @boundary
    def get_last_self_attention(self, x, *, key=None):                 # reference :275-292
        if not self.inference:
            raise ValueError("Model being evaluated outside inference mode. Try in inference mode.")
        x = self._tokens(x)
        for blk in self.blocks[:-1]:
            x = blk(x)
        return self.blocks[-1](x, return_attention=True)


def _vit(patch_size, embed_dim, depth, num_heads, mlp_ratio, torch_weights, key, kwargs):
    model = VisionTransformer(patch_size=patch_size, embed_dim=embed_dim, depth=depth, num_heads=num_heads,
                              mlp_ratio=mlp_ratio, key=key, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


def vit_tiny(patch_size=16, embed_dim=192, depth=12, num_heads=3, mlp_ratio=4, torch_weights: str = None, *,
             key=None, **kwargs):
    return _vit(patch_size, embed_dim, depth, num_heads, mlp_ratio, torch_weights, key, kwargs)


def vit_small(patch_size=16, embed_dim=384, depth=12, num_heads=6, mlp_ratio=4, torch_weights: str = None, *,
              key=None, **kwargs):
    return _vit(patch_size, embed_dim, depth, num_heads, mlp_ratio, torch_weights, key, kwargs)


def vit_base(patch_size=16, embed_dim=768, depth=12, num_heads=12, mlp_ratio=4, torch_weights: str = None, *,
             key=None, **kwargs):
    """ViT-B/16 (reference :370-404): 12 blocks, 768-d, 12 heads, 17.6 GMAC per 224x224 image."""
    return _vit(patch_size, embed_dim, depth, num_heads, mlp_ratio, torch_weights, key, kwargs)


vit_b_16 = vit_base   # the north star's name for vit_base(patch_size=16)
