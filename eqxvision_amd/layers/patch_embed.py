"""Image -> patch tokens (reference layers/patch_embed.py:11-84).

The reference runs Conv2d(k = s = patch), ravels each channel and moves the channel axis last.  Here
the same contraction is one implicit-GEMM launch that reads the NCHW image directly (no im2col
buffer) and writes token rows [P, D] -- the layout every following Linear wants."""
from __future__ import annotations

from typing import Optional, Tuple, Union

from .. import nn, ops
from .. import random as jr
from .._module import Module
from ..nn import boundary


class PatchEmbed(Module):
    img_size: Tuple[int]
    patch_size: Tuple[int]
    grid_size: Tuple[int]
    num_patches: int
    flatten: bool
    proj: nn.Conv2d
    norm: Module

    def __init__(self, img_size: Union[int, Tuple[int]] = 224, patch_size: Union[int, Tuple[int]] = 16,
                 in_chans: int = 3, embed_dim: int = 768, norm_layer=None, flatten: bool = True, *, key=None):
        self.img_size = img_size if isinstance(img_size, tuple) else (img_size, img_size)
        self.patch_size = patch_size if isinstance(patch_size, tuple) else (patch_size, patch_size)
        self.grid_size = (self.img_size[0] // self.patch_size[0], self.img_size[1] // self.patch_size[1])
        self.num_patches = self.grid_size[0] * self.grid_size[1]
        self.flatten = flatten
        if key is None:
            key = jr.PRNGKey(0)
        self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size, stride=patch_size, key=key)
        self.norm = norm_layer(embed_dim) if norm_layer else nn.Identity()

    def _check(self, x):
        shp = tuple(x.shape)
        if len(shp) != 3:
            raise ValueError(f"PatchEmbed expects (in_chans, H, W), got {shp}")
        C, H, W = shp
        if H != self.img_size[0] or W != self.img_size[1]:           # reference :74-77
            raise ValueError(f"Input image height ({H},{W}) doesn't match model ({self.img_size}).")

    def __call__(self, x, *, key=None):
        self._check(x)            # shape error first, like the reference (:74-77) -- no device needed
        return self._forward(x)

    @boundary
    def _forward(self, x):
        if self.flatten and x.kind == "img":
            t = ops.patch_embed_tokens(x, self.proj, None, None, 0)
        else:
            t = ops.conv2d(x, self.proj)
            if self.flatten:                                          # map [B,gh,gw,D] == rows [B,P,D]
                from .._act import Act
                B, gh, gw, D = t.t.shape
                t = Act(t.t.reshape(B, gh * gw, D), "seq", t.batched)
        if isinstance(self.norm, nn.LayerNorm) and self.flatten:
            # reference patch_embed.py:82-83 calls eqx.nn.LayerNorm(embed_dim) on the whole (P, D) array: the equinox the
            # reference targets normalises over ALL P*D elements, newer ones reject the shape, timm normalises per token.
            # Three different answers -- refuse instead of silently picking one (no hot config uses it: vit.py:224-228).
            raise NotImplementedError("PatchEmbed(norm_layer=LayerNorm) on flattened tokens is ambiguous across equinox versions; "
                                      "apply the LayerNorm to the returned tokens explicitly")
        return self.norm(t)
