"""Squeeze-and-excitation (reference layers/squeeze.py:11-61): global mean -> 1x1 conv -> activation -> 1x1 conv -> scale
activation -> x * scale.  Same fields (`avgpool`, `fc1`, `fc2`, `activation`, `scale_activation`) and constructor.

On the device the squeeze path works on a [B, 1, 1, C] map (the two 1x1 convolutions are B-row GEMMs) and the excitation is one
broadcast-multiply pass over the feature map."""
from __future__ import annotations

from typing import Callable

from .. import nn, ops
from .. import random as jr
from .._module import Module
from ..nn import boundary


class SqueezeExcitation(Module):
    avgpool: nn.AdaptiveAvgPool2d
    fc1: nn.Conv2d
    fc2: nn.Conv2d
    activation: nn.Lambda
    scale_activation: nn.Lambda

    def __init__(self, input_channels: int, squeeze_channels: int, activation: Callable = None, scale_activation: Callable = None,
                 *, key=None) -> None:
        k1, k2 = jr.split(key if key is not None else jr.PRNGKey(0), 2)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.fc1 = nn.Conv2d(input_channels, squeeze_channels, 1, key=k1)
        self.fc2 = nn.Conv2d(squeeze_channels, input_channels, 1, key=k2)
        self.activation = nn.Lambda(activation if activation is not None else nn.relu)
        self.scale_activation = nn.Lambda(scale_activation if scale_activation is not None else nn.sigmoid)

    @boundary
    def __call__(self, x, *, key=None):
        x = ops.as_map(x)
        a, g = nn.act_name(self.activation.fn), nn.act_name(self.scale_activation.fn)
        if a and g and type(self.avgpool) is nn.AdaptiveAvgPool2d and tuple(self.avgpool.target_shape) == (1, 1) \
                and type(self.fc1) is nn.Conv2d and type(self.fc2) is nn.Conv2d:
            s = ops.se_scale(x, self.fc1, self.fc2, a, g)            # the whole squeeze path in one launch
            if s is not None:
                return ops.channel_scale(x, s)
        s = self.avgpool(x)
        a = nn.act_name(self.activation.fn)
        s = ops.conv2d(s, self.fc1, None, a) if a else self.activation(ops.conv2d(s, self.fc1))
        g = nn.act_name(self.scale_activation.fn)
        s = ops.conv2d(s, self.fc2, None, g) if g else self.scale_activation(ops.conv2d(s, self.fc2))
        return ops.channel_scale(x, s)
