"""Channel-wise LayerNorm / Linear on (C,H,W) inputs (reference layers/extensions_2d.py:9-50).

The reference transposes CHW -> (HW, C), vmaps the 1-D layer and transposes back.  On the device a
feature map is already pixel-major / channel-contiguous (NHWC), so both are plain row kernels on
the map's rows with no data movement."""
from __future__ import annotations

from .. import nn, ops
from ..nn import boundary


class LayerNorm2d(nn.LayerNorm):
    @boundary
    def __call__(self, x, *, key=None):
        if len(x.shape) != 3:
            raise ValueError(f"LayerNorm2d expects (channels, dim_0, dim_1), got {x.shape}")
        return ops.layernorm(ops.as_map(x), self)


class Linear2d(nn.Linear):
    @boundary
    def __call__(self, x, *, key=None):
        if len(x.shape) != 3:
            raise ValueError(f"Linear2d expects (channels, dim_0, dim_1), got {x.shape}")
        return ops.linear(ops.as_map(x), self)
