"""Lite R-ASPP on MobileNetV3-Large (reference models/segmentation/lraspp.py:13-175): the backbone is the `features` stack of
`mobilenet_v3_large(dilated=True)` wrapped by `intermediate_layer_getter` at indices [4, 16] (a stride-8 map with 40 channels and
the stride-16 map with 960); the head gates a 1x1-conv branch of the deep map with a globally pooled sigmoid branch, up-samples it
to the shallow map and adds the two 1x1 classifiers; the result is resized to the input resolution."""
from __future__ import annotations

from typing import Callable, Optional

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...experimental import intermediate_layer_getter
from ...nn import boundary
from ...utils import load_torch_weights
from ..classification.mobilenetv3 import mobilenet_v3_large


class LRASPPHead(Module):
    cbr: Module
    scale: Module
    low_classifier: Module
    high_classifier: Module

    def __init__(self, low_channels: int, high_channels: int, num_classes: int, inter_channels: int, key=None) -> None:
        k_cbr, k_scale, k_low, k_high = jr.split(key if key is not None else jr.PRNGKey(0), 4)
        self.cbr = nn.Sequential([nn.Conv2d(high_channels, inter_channels, 1, use_bias=False, key=k_cbr),
                                  nn.BatchNorm(inter_channels, axis_name="batch"), nn.Lambda(nn.relu)])
        self.scale = nn.Sequential([nn.AdaptiveAvgPool2d(1), nn.Conv2d(high_channels, inter_channels, 1, use_bias=False, key=k_scale),
                                    nn.Lambda(nn.sigmoid)])
        self.low_classifier = nn.Conv2d(low_channels, num_classes, 1, key=k_low)
        self.high_classifier = nn.Conv2d(inter_channels, num_classes, 1, key=k_high)

    def __call__(self, x, *, key=None):
        low, high = ops.as_map(x[0]), ops.as_map(x[1])
        y = ops.channel_scale(self.cbr(high), self.scale(high))              # x * s, s one gate per channel
        y = ops.resize_bilinear(y, tuple(low.shape[-2:]))
        return ops.add(ops.conv2d(low, self.low_classifier), ops.conv2d(y, self.high_classifier))


class LRASPP(Module):
    backbone: Module
    classifier: Module

    def __init__(self, backbone: Module, low_channels: int, high_channels: int, num_classes: int, inter_channels: int = 128,
                 key=None) -> None:
        self.backbone = backbone
        self.classifier = LRASPPHead(low_channels, high_channels, num_classes, inter_channels, key=key)

    @boundary
    def __call__(self, x, *, key=None):
        size = tuple(x.shape[-2:])
        _, features = self.backbone(x)
        return None, ops.resize_bilinear(self.classifier(features), size, final=True)


def lraspp_mobilenet_v3_large(num_classes: Optional[int] = 21, backbone: Module = None, intermediate_layers: Callable = None,
                              torch_weights: str = None, *, key=None) -> LRASPP:
    """Sample call (reference docstring): `lraspp_mobilenet_v3_large(backbone=mobilenet_v3_large(dilated=True),
    intermediate_layers=lambda x: [4, 16], torch_weights=SEGMENTATION_URLS['lraspp_mobilenetv3_large'])`."""
    if key is None:
        key = jr.PRNGKey(0)
    if num_classes is None:
        num_classes = 21
    if backbone is None:
        backbone = mobilenet_v3_large(dilated=True)
    if intermediate_layers is None:
        intermediate_layers = lambda m: [4, 16]
    stack = backbone.features
    low_c, high_c = (stack.layers[i].out_channels for i in intermediate_layers(stack))
    model = LRASPP(intermediate_layer_getter(stack, intermediate_layers), low_c, high_c, num_classes=num_classes, key=key)
    if torch_weights:
        return load_torch_weights(model, torch_weights)
    return model
