"""Conv -> Norm -> Activation block (reference layers/conv_norm_activation.py:10-86).

Because it is an `nn.Sequential` of [Conv2d, BatchNorm, Lambda(act)], calling it on the device
hits `nn.Sequential`'s peephole and becomes ONE implicit-GEMM launch with the BatchNorm folded
into the fp32 scale/shift epilogue and the activation applied before the bf16 store.
"""
from __future__ import annotations

from functools import partial
from typing import Callable, Optional

from .. import nn
from .. import random as jr


class ConvNormActivation(nn.Sequential):
    out_channels: int

    def __init__(self, in_channels: int, out_channels: int, kernel_size: int = 3, stride: int = 1,
                 padding: Optional[int] = None, groups: int = 1, norm_layer: Optional[Callable] = nn.BatchNorm,
                 activation_layer: Optional[Callable] = nn.relu, dilation: int = 1,
                 use_bias: Optional[bool] = None, *, key=None) -> None:
        if key is None:
            key = jr.PRNGKey(0)
        if padding is None:                                  # reference :56-57
            padding = (kernel_size - 1) // 2 * dilation
        if use_bias is None:                                 # reference :58-59
            use_bias = norm_layer is None
        stack = [nn.Conv2d(in_channels, out_channels, kernel_size, stride, padding, dilation=dilation,
                           groups=groups, use_bias=use_bias, key=key)]
        if norm_layer is not None:
            base = norm_layer.func if isinstance(norm_layer, partial) else norm_layer
            if base is nn.BatchNorm:                         # reference :73-80: BN gets axis_name="batch"
                stack.append(norm_layer(out_channels, axis_name="batch"))
            else:
                stack.append(norm_layer(out_channels))
        if activation_layer is not None:
            stack.append(nn.Lambda(activation_layer))
        super().__init__(stack)
        self.out_channels = out_channels
