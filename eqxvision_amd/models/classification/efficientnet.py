"""EfficientNet B0-B7 and EfficientNetV2 S / M / L (reference models/classification/efficientnet.py:19-715; SURVEY section 8
row f1): same config classes (`_MBConvConfig`, `_FusedMBConvConfig`), blocks, fields (`features`, `avgpool`, `classifier`) and
constructors.  Stochastic depth is the identity in inference (`layers.DropPath`).

Device lowering: MBConv = 1x1 expansion GEMM (+BN; SiLU as an element-wise pass), k x k depthwise (+BN+SiLU inside the depthwise
kernel), squeeze-excitation (B-row GEMMs + one broadcast multiply), 1x1 projection GEMM (+BN, + the block input in its epilogue);
FusedMBConv = a regular k x k convolution on the matrix cores instead of expansion + depthwise."""
from __future__ import annotations

import copy
import math
from functools import partial
from typing import Any, Callable, List, Optional, Sequence, Tuple, Union

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...layers import ConvNormActivation, DropPath, SqueezeExcitation
from ...nn import boundary
from ...utils import _make_divisible, load_torch_weights


class _MBConvConfigData:
    """(expand_ratio, kernel, stride, input_channels, out_channels, num_layers, block) -- a row of Table 1 (EfficientNet) /
    Table 4 (EfficientNetV2); a plain class instead of the reference's dataclass, same attributes."""

    def __init__(self, expand_ratio: float, kernel: int, stride: int, input_channels: int, out_channels: int, num_layers: int,
                 block: Callable):
        self.expand_ratio, self.kernel, self.stride = expand_ratio, kernel, stride
        self.input_channels, self.out_channels, self.num_layers, self.block = input_channels, out_channels, num_layers, block

    @staticmethod
    def adjust_channels(channels: int, width_mult: float, min_value: Optional[int] = None) -> int:
        return _make_divisible(channels * width_mult, 8, min_value)


class _MBConvConfig(_MBConvConfigData):
    def __init__(self, expand_ratio: float, kernel: int, stride: int, input_channels: int, out_channels: int, num_layers: int,
                 width_mult: float = 1.0, depth_mult: float = 1.0, block: Optional[Callable] = None) -> None:
        super().__init__(expand_ratio, kernel, stride, self.adjust_channels(input_channels, width_mult),
                         self.adjust_channels(out_channels, width_mult), self.adjust_depth(num_layers, depth_mult),
                         block if block is not None else _MBConv)

    @staticmethod
    def adjust_depth(num_layers: int, depth_mult: float):
        return int(math.ceil(num_layers * depth_mult))


class _FusedMBConvConfig(_MBConvConfigData):
    def __init__(self, expand_ratio: float, kernel: int, stride: int, input_channels: int, out_channels: int, num_layers: int,
                 block: Optional[Callable] = None) -> None:
        super().__init__(expand_ratio, kernel, stride, input_channels, out_channels, num_layers,
                         block if block is not None else _FusedMBConv)


def _residual_tail(block: nn.Sequential, x, key):
    """block(x) + x with the add inside the last projection's GEMM epilogue when that layer is [Conv2d, BatchNorm]."""
    last = block.layers[-1]
    L = getattr(last, "layers", None)
    if L is not None and len(L) == 2 and type(L[0]) is nn.Conv2d and isinstance(L[1], nn.BatchNorm) and L[1].inference:
        x = ops.as_map(x)
        h = block[:-1](x, key=key)
        return ops.conv2d(h, L[0], L[1], None, residual=x)
    return ops.add(block(x, key=key), x)


class _MBConv(Module):
    use_res_connect: bool
    block: nn.Sequential
    stochastic_depth: DropPath
    out_channels: int

    def __init__(self, cnf: _MBConvConfig, stochastic_depth_prob: float, norm_layer: Callable,
                 se_layer: Callable = SqueezeExcitation, *, key=None) -> None:
        if not (1 <= cnf.stride <= 2):
            raise ValueError("illegal stride value")
        k_expand, k_dw, k_se, k_project = jr.split(key if key is not None else jr.PRNGKey(0), 4)
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.silu
        expanded = cnf.adjust_channels(cnf.input_channels, cnf.expand_ratio)
        stack: List[Module] = []
        if expanded != cnf.input_channels:
            stack.append(ConvNormActivation(cnf.input_channels, expanded, kernel_size=1, norm_layer=norm_layer, activation_layer=act,
                                            key=k_expand))
        stack.append(ConvNormActivation(expanded, expanded, kernel_size=cnf.kernel, stride=cnf.stride, groups=expanded,
                                        norm_layer=norm_layer, activation_layer=act, key=k_dw))
        stack.append(se_layer(expanded, max(1, cnf.input_channels // 4), activation=act, key=k_se))
        stack.append(ConvNormActivation(expanded, cnf.out_channels, kernel_size=1, norm_layer=norm_layer, activation_layer=None,
                                        key=k_project))
        self.block = nn.Sequential(stack)
        self.stochastic_depth = DropPath(stochastic_depth_prob, mode="per_channel")
        self.out_channels = cnf.out_channels

    @boundary
    def __call__(self, x, *, key=None):
        if not self.use_res_connect:
            return self.block(x, key=key)
        sd = self.stochastic_depth
        if sd.inference or sd.p == 0.0:
            return _residual_tail(self.block, x, key)
        keys = [None, None] if key is None else jr.split(key, 2)          # reference :181-184 / :261-264
        return ops.add(sd(self.block(x, key=keys[0]), key=keys[1]), x)


class _FusedMBConv(Module):
    use_res_connect: bool
    block: nn.Sequential
    stochastic_depth: DropPath
    out_channels: int

    def __init__(self, cnf: _FusedMBConvConfig, stochastic_depth_prob: float, norm_layer: Callable, *, key=None) -> None:
        if not (1 <= cnf.stride <= 2):
            raise ValueError("illegal stride value")
        k_expand, k_project, k_single = jr.split(key if key is not None else jr.PRNGKey(0), 3)
        self.use_res_connect = cnf.stride == 1 and cnf.input_channels == cnf.out_channels
        act = nn.silu
        expanded = cnf.adjust_channels(cnf.input_channels, cnf.expand_ratio)
        if expanded != cnf.input_channels:
            stack = [ConvNormActivation(cnf.input_channels, expanded, kernel_size=cnf.kernel, stride=cnf.stride, norm_layer=norm_layer,
                                        activation_layer=act, key=k_expand),
                     ConvNormActivation(expanded, cnf.out_channels, kernel_size=1, norm_layer=norm_layer, activation_layer=None,
                                        key=k_project)]
        else:
            stack = [ConvNormActivation(cnf.input_channels, cnf.out_channels, kernel_size=cnf.kernel, stride=cnf.stride,
                                        norm_layer=norm_layer, activation_layer=act, key=k_single)]
        self.block = nn.Sequential(stack)
        self.stochastic_depth = DropPath(stochastic_depth_prob, mode="local")
        self.out_channels = cnf.out_channels

    @boundary
    def __call__(self, x, *, key=None):
        if not self.use_res_connect:
            return self.block(x, key=key)
        sd = self.stochastic_depth
        if sd.inference or sd.p == 0.0:
            return _residual_tail(self.block, x, key)
        keys = [None, None] if key is None else jr.split(key, 2)          # reference :181-184 / :261-264
        return ops.add(sd(self.block(x, key=keys[0]), key=keys[1]), x)


class EfficientNet(Module):
    features: nn.Sequential
    avgpool: nn.AdaptiveAvgPool2d
    classifier: nn.Sequential

    def __init__(self, inverted_residual_setting: Sequence[Union[_MBConvConfig, _FusedMBConvConfig]], dropout: float,
                 stochastic_depth_prob: float = 0.2, num_classes: int = 1000, norm_layer: Optional[Callable] = None,
                 last_channel: Optional[int] = None, *, key=None) -> None:
        if not inverted_residual_setting:
            raise ValueError("The inverted_residual_setting should not be empty")
        if not (isinstance(inverted_residual_setting, Sequence)
                and all(isinstance(s, _MBConvConfigData) for s in inverted_residual_setting)):
            raise TypeError("The inverted_residual_setting should be List[MBConvConfig]")
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, 3)
        if norm_layer is None:
            norm_layer = nn.BatchNorm
        stack: List[Module] = [ConvNormActivation(3, inverted_residual_setting[0].input_channels, kernel_size=3, stride=2,
                                                  norm_layer=norm_layer, activation_layer=nn.silu, key=keys[0])]
        total = sum(cnf.num_layers for cnf in inverted_residual_setting)
        block_id = 0
        for cnf in inverted_residual_setting:
            stage: List[Module] = []
            for _ in range(cnf.num_layers):
                keys = jr.split(keys[1], 2)
                bc = copy.copy(cnf)                       # later blocks of a stage: stride 1, in = out (reference :347-352)
                if stage:
                    bc.input_channels = bc.out_channels
                    bc.stride = 1
                stage.append(bc.block(bc, stochastic_depth_prob * float(block_id) / total, norm_layer, key=keys[0]))
                block_id += 1
            stack.append(nn.Sequential(stage))
        keys = jr.split(keys[1], 2)
        tail_in = inverted_residual_setting[-1].out_channels
        tail_out = last_channel if last_channel is not None else 4 * tail_in
        stack.append(ConvNormActivation(tail_in, tail_out, kernel_size=1, norm_layer=norm_layer, activation_layer=nn.silu, key=keys[0]))
        self.features = nn.Sequential(stack)
        self.avgpool = nn.AdaptiveAvgPool2d(1)
        self.classifier = nn.Sequential([nn.Dropout(p=dropout), nn.Linear(tail_out, num_classes, key=keys[1])])

    def __call__(self, x, *, key):
        if key is None:                                  # the reference splits the key first thing (:392)
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x, key)

    @boundary
    def _forward(self, x, key=None):
        from ..._act import head_fp32
        from ...transforms import _needs_eager
        if key is not None and _needs_eager(self):       # training mode: stochastic depth in the blocks, the classifier's Dropout
            keys = jr.split(key, 2)                      # reference :398-403
            x = self.features(x, key=keys[0])
            return self.classifier(ops.flatten(self.avgpool(x)), key=keys[1])
        x = self.features(x)
        if type(self.avgpool) is nn.AdaptiveAvgPool2d and head_fp32():
            x = ops.adaptive_avgpool2d(x, self.avgpool.target_shape, out_fp32=True)
        else:
            x = self.avgpool(x)
        x = ops.flatten(x)
        head = self.classifier.layers[-1]
        if type(head) is nn.Linear:
            x = self.classifier[:-1](x)
            return ops.linear_head(x, head)
        return self.classifier(x)


def _efficientnet(arch: str, inverted_residual_setting, dropout: float, last_channel: Optional[int], torch_weights: str,
                  **kwargs: Any) -> EfficientNet:
    model = EfficientNet(inverted_residual_setting, dropout, last_channel=last_channel, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


_B_TABLE = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
            (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))
# (fused?, expand, kernel, stride, in, out, layers)
_V2_TABLES = {
    "efficientnet_v2_s": ((1, 1, 3, 1, 24, 24, 2), (1, 4, 3, 2, 24, 48, 4), (1, 4, 3, 2, 48, 64, 4), (0, 4, 3, 2, 64, 128, 6),
                          (0, 6, 3, 1, 128, 160, 9), (0, 6, 3, 2, 160, 256, 15)),
    "efficientnet_v2_m": ((1, 1, 3, 1, 24, 24, 3), (1, 4, 3, 2, 24, 48, 5), (1, 4, 3, 2, 48, 80, 5), (0, 4, 3, 2, 80, 160, 7),
                          (0, 6, 3, 1, 160, 176, 14), (0, 6, 3, 2, 176, 304, 18), (0, 6, 3, 1, 304, 512, 5)),
    "efficientnet_v2_l": ((1, 1, 3, 1, 32, 32, 4), (1, 4, 3, 2, 32, 64, 7), (1, 4, 3, 2, 64, 96, 7), (0, 4, 3, 2, 96, 192, 10),
                          (0, 6, 3, 1, 192, 224, 19), (0, 6, 3, 2, 224, 384, 25), (0, 6, 3, 1, 384, 640, 7)),
}


def _efficientnet_conf(arch: str, **kwargs: Any) -> Tuple[Sequence[_MBConvConfigData], Optional[int]]:
    if arch.startswith("efficientnet_b"):
        row = partial(_MBConvConfig, width_mult=kwargs.pop("width_mult"), depth_mult=kwargs.pop("depth_mult"))
        return [row(*r) for r in _B_TABLE], None
    for name, table in _V2_TABLES.items():
        if arch.startswith(name):
            return [(_FusedMBConvConfig if r[0] else _MBConvConfig)(*r[1:]) for r in table], 1280
    raise ValueError(f"Unsupported model type {arch}")


# name -> (width_mult, depth_mult, dropout, BatchNorm override)      reference :478-715
_B_VARIANTS = {
    "efficientnet_b0": (1.0, 1.0, 0.2, None), "efficientnet_b1": (1.0, 1.1, 0.2, None), "efficientnet_b2": (1.1, 1.2, 0.3, None),
    "efficientnet_b3": (1.2, 1.4, 0.3, None), "efficientnet_b4": (1.4, 1.8, 0.4, None),
    "efficientnet_b5": (1.6, 2.2, 0.4, dict(eps=0.001, momentum=0.01)), "efficientnet_b6": (1.8, 2.6, 0.5, dict(eps=0.001, momentum=0.01)),
    "efficientnet_b7": (2.0, 3.1, 0.5, dict(eps=0.001, momentum=0.01)),
}
_V2_DROPOUT = {"efficientnet_v2_s": 0.2, "efficientnet_v2_m": 0.3, "efficientnet_v2_l": 0.4}


def _variant(name: str):
    def make(torch_weights: str = None, **kwargs: Any) -> EfficientNet:
        if name in _B_VARIANTS:
            wm, dm, drop, bn = _B_VARIANTS[name]
            setting, last = _efficientnet_conf(name, width_mult=wm, depth_mult=dm)
            if bn is not None:
                kwargs = dict(kwargs, norm_layer=partial(nn.BatchNorm, **bn))
        else:
            drop = _V2_DROPOUT[name]
            setting, last = _efficientnet_conf(name)
            kwargs = dict(kwargs, norm_layer=partial(nn.BatchNorm, eps=1e-03))
        return _efficientnet(name, setting, drop, last, torch_weights, **kwargs)

    make.__name__ = make.__qualname__ = name
    make.__doc__ = f"{name} (reference models/classification/efficientnet.py); `torch_weights`: torchvision checkpoint path / URL."
    return make


efficientnet_b0, efficientnet_b1, efficientnet_b2, efficientnet_b3 = (_variant(f"efficientnet_b{i}") for i in range(4))
efficientnet_b4, efficientnet_b5, efficientnet_b6, efficientnet_b7 = (_variant(f"efficientnet_b{i}") for i in range(4, 8))
efficientnet_v2_s, efficientnet_v2_m, efficientnet_v2_l = (_variant(f"efficientnet_v2_{s}") for s in "sml")
