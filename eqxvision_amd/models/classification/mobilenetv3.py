"""MobileNetV3 large / small (reference models/classification/mobilenetv3.py:16-389): same
`_InvertedResidualConfig` tables, fields (`features`, `avgpool`, `classifier`) and constructors (incl. `dilated=` for the
segmentation backbone and the BatchNorm eps of 1e-3).

Device lowering of a block: 1x1 expansion (MFMA GEMM; relu fused, hard_swish as one element-wise pass), k x k depthwise
(+BN + activation in the depthwise kernel itself), squeeze-excitation (B-row GEMMs on the pooled vector + one broadcast
multiply), 1x1 projection (+BN, + the block input in the GEMM epilogue when `use_res_connect`)."""
from __future__ import annotations

from functools import partial
from typing import Any, Callable, List, Optional, Sequence

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...layers import ConvNormActivation
from ...layers import SqueezeExcitation as SElayer
from ...nn import boundary
from ...utils import _make_divisible, load_torch_weights


class _InvertedResidualConfig:
    """One row of Tables 1 / 2 of the MobileNetV3 paper (reference :16-43)."""

    def __init__(self, input_channels: int, kernel: int, expanded_channels: int, out_channels: int, use_se: bool,
                 activation: str, stride: int, dilation: int, width_mult: float):
        self.input_channels = self.adjust_channels(input_channels, width_mult)
        self.kernel = kernel
        self.expanded_channels = self.adjust_channels(expanded_channels, width_mult)
        self.out_channels = self.adjust_channels(out_channels, width_mult)
        self.use_se = use_se
        self.use_hs = activation == "HS"
        self.stride = stride
        self.dilation = dilation

    @staticmethod
    def adjust_channels(channels: int, width_mult: float):
        return _make_divisible(channels * width_mult, 8)


class _InvertedResidual(Module):
    use_res_connect: int
    block: nn.Sequential
    out_channels: int

    def __init__(self, cnf: _InvertedResidualConfig, norm_layer: Callable,
                 se_layer: Callable = partial(SElayer, scale_activation=nn.hard_sigmoid), *, key=None):
        k_expand, k_dw, k_se, k_project = jr.split(key if key is not None else jr.PRNGKey(0), 4)
        if not (1 <= cnf.stride <= 2):
            raise ValueError("illegal stride value")
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.hard_swish if cnf.use_hs else nn.relu
        stack: List[Module] = []
        if cnf.expanded_channels != cnf.input_channels:
            stack.append(ConvNormActivation(cnf.input_channels, cnf.expanded_channels, kernel_size=1, norm_layer=norm_layer,
                                            activation_layer=act, key=k_expand))
        stack.append(ConvNormActivation(cnf.expanded_channels, cnf.expanded_channels, kernel_size=cnf.kernel,
                                        stride=1 if cnf.dilation > 1 else cnf.stride, dilation=cnf.dilation,
                                        groups=cnf.expanded_channels, norm_layer=norm_layer, activation_layer=act, key=k_dw))
        if cnf.use_se:
            stack.append(se_layer(cnf.expanded_channels, _make_divisible(cnf.expanded_channels // 4, 8), key=k_se))
        stack.append(ConvNormActivation(cnf.expanded_channels, cnf.out_channels, kernel_size=1, norm_layer=norm_layer,
                                        activation_layer=None, key=k_project))
        self.block = nn.Sequential(stack)
        self.out_channels = cnf.out_channels

    @boundary
    def __call__(self, x, *, key=None):
        if not self.use_res_connect:
            return self.block(x, key=key)
        last = self.block.layers[-1]
        L = getattr(last, "layers", None)
        if L is not None and len(L) == 2 and type(L[0]) is nn.Conv2d and isinstance(L[1], nn.BatchNorm) and L[1].inference:
            x = ops.as_map(x)
            h = self.block[:-1](x, key=key)
            return ops.conv2d(h, L[0], L[1], None, residual=x)        # result += x in the projection's epilogue
        return ops.add(self.block(x, key=key), x)


class MobileNetV3(Module):
    features: nn.Sequential
    avgpool: nn.AdaptiveAvgPool2d
    classifier: nn.Sequential

    def __init__(self, inverted_residual_setting: List[_InvertedResidualConfig], last_channel: int, num_classes: int = 1000,
                 block: Optional[Callable] = None, norm_layer: Optional[Callable] = None, dropout: float = 0.2, *, key=None) -> None:
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, 5)
        if not inverted_residual_setting:
            raise ValueError("The inverted_residual_setting should not be empty")
        if not (isinstance(inverted_residual_setting, Sequence)
                and all(isinstance(s, _InvertedResidualConfig) for s in inverted_residual_setting)):
            raise TypeError("The inverted_residual_setting should be List[InvertedResidualConfig]")
        block = block or _InvertedResidual
        if norm_layer is None:
            norm_layer = partial(nn.BatchNorm, eps=0.001, momentum=0.01)
        first = inverted_residual_setting[0].input_channels
        stack: List[Module] = [ConvNormActivation(3, first, kernel_size=3, stride=2, norm_layer=norm_layer,
                                                  activation_layer=nn.hard_swish, key=keys[0])]
        for cnf in inverted_residual_setting:
            stack.append(block(cnf, norm_layer, key=keys[1]))          # the reference hands every block the same key (:196)
        tail_in = inverted_residual_setting[-1].out_channels
        tail_out = 6 * tail_in
        stack.append(ConvNormActivation(tail_in, tail_out, kernel_size=1, norm_layer=norm_layer, activation_layer=nn.hard_swish,
                                        key=keys[2]))
        self.features = nn.Sequential(stack)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential([nn.Linear(tail_out, last_channel, key=keys[3]), nn.Lambda(nn.hard_swish),
                                         nn.Dropout(p=dropout), nn.Linear(last_channel, num_classes, key=keys[4])])

    def __call__(self, x, *, key):
        if key is None:                                  # the reference splits the key first thing (:240)
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x, key)

    @boundary
    def _forward(self, x, key=None):
        from ...transforms import _needs_eager
        if key is not None and _needs_eager(self):       # training mode: the classifier's Dropout draws from keys[2]
            keys = jr.split(key, 3)                      # reference mobilenetv3.py:242-246
            x = self.features(x, key=keys[0])
            return self.classifier(ops.flatten(self.avgpool(x)), key=keys[2])
        x = self.features(x)
        x = self.avgpool(x)
        x = ops.flatten(x)
        head = self.classifier.layers[-1]
        if type(head) is nn.Linear:                      # keep the logits in fp32
            x = self.classifier[:-1](x)
            return ops.linear(x, head, out_fp32=True)
        return self.classifier(x)


def _mobilenet_v3_conf(arch: str, width_mult: float = 1.0, reduced_tail: bool = False, dilated: bool = False, **kwargs: Any):
    """The two published tables (reference :248-341): (in, kernel, expanded, out, SE, activation, stride, dilation)."""
    div = 2 if reduced_tail else 1
    dil = 2 if dilated else 1
    row = partial(_InvertedResidualConfig, width_mult=width_mult)
    adjust = partial(_InvertedResidualConfig.adjust_channels, width_mult=width_mult)
    if arch == "mobilenet_v3_large":
        table = [(16, 3, 16, 16, False, "RE", 1, 1), (16, 3, 64, 24, False, "RE", 2, 1), (24, 3, 72, 24, False, "RE", 1, 1),
                 (24, 5, 72, 40, True, "RE", 2, 1), (40, 5, 120, 40, True, "RE", 1, 1), (40, 5, 120, 40, True, "RE", 1, 1),
                 (40, 3, 240, 80, False, "HS", 2, 1), (80, 3, 200, 80, False, "HS", 1, 1), (80, 3, 184, 80, False, "HS", 1, 1),
                 (80, 3, 184, 80, False, "HS", 1, 1), (80, 3, 480, 112, True, "HS", 1, 1), (112, 3, 672, 112, True, "HS", 1, 1),
                 (112, 5, 672, 160 // div, True, "HS", 2, dil), (160 // div, 5, 960 // div, 160 // div, True, "HS", 1, dil),
                 (160 // div, 5, 960 // div, 160 // div, True, "HS", 1, dil)]
        last_channel = adjust(1280 // div)
    elif arch == "mobilenet_v3_small":
        table = [(16, 3, 16, 16, True, "RE", 2, 1), (16, 3, 72, 24, False, "RE", 2, 1), (24, 3, 88, 24, False, "RE", 1, 1),
                 (24, 5, 96, 40, True, "HS", 2, 1), (40, 5, 240, 40, True, "HS", 1, 1), (40, 5, 240, 40, True, "HS", 1, 1),
                 (40, 5, 120, 48, True, "HS", 1, 1), (48, 5, 144, 48, True, "HS", 1, 1),
                 (48, 5, 288, 96 // div, True, "HS", 2, dil), (96 // div, 5, 576 // div, 96 // div, True, "HS", 1, dil),
                 (96 // div, 5, 576 // div, 96 // div, True, "HS", 1, dil)]
        last_channel = adjust(1024 // div)
    else:
        raise ValueError(f"Unsupported model type {arch}")
    return [row(*r) for r in table], last_channel


def _mobilenet_v3(arch: str, inverted_residual_setting: List[_InvertedResidualConfig], last_channel: int, **kwargs: Any):
    return MobileNetV3(inverted_residual_setting, last_channel, **kwargs)


def mobilenet_v3_large(torch_weights: str = None, **kwargs: Any) -> MobileNetV3:
    arch = "mobilenet_v3_large"
    dilated = kwargs.pop("dilated", False)
    setting, last_channel = _mobilenet_v3_conf(arch, dilated=dilated, **kwargs)
    model = _mobilenet_v3(arch, setting, last_channel, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


def mobilenet_v3_small(torch_weights: str = None, **kwargs: Any) -> MobileNetV3:
    arch = "mobilenet_v3_small"
    setting, last_channel = _mobilenet_v3_conf(arch, **kwargs)
    model = _mobilenet_v3(arch, setting, last_channel, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model
