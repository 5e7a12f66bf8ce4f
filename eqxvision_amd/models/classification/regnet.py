"""RegNet X / Y (reference models/classification/regnet.py:15-676): `SimpleStemIN`, `BottleneckTransform`,
`ResBottleneckBlock`, `AnyStage`, `BlockParams.from_init_params` (the quantised-linear width rule) and the 15 published
configurations, same fields (`stem`, `trunk_output`, `avgpool`, `fc`).

Device lowering of a block: 1x1 conv (+BN+relu) GEMM, 3x3 GROUPED conv (+BN+relu) on the matrix cores through per-tile input
windows (`mv_conv2d_nhwc_grouped64_fwd`; group widths 8 ... 264), squeeze-excitation for the Y variants, 1x1 conv (+BN) GEMM with
the (projected) shortcut and the relu in its epilogue."""
from __future__ import annotations

from functools import partial
from typing import Any, Callable, List, Optional, Tuple

import numpy as np

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...layers import ConvNormActivation, SqueezeExcitation
from ...nn import boundary
from ...utils import _make_divisible, load_torch_weights


class SimpleStemIN(ConvNormActivation):
    def __init__(self, width_in: int, width_out: int, norm_layer: Optional[Callable], activation_layer: Optional[Callable], *,
                 key=None) -> None:
        super().__init__(width_in, width_out, kernel_size=3, stride=2, norm_layer=norm_layer, activation_layer=activation_layer, key=key)


class BottleneckTransform(nn.Sequential):
    def __init__(self, width_in: int, width_out: int, stride: int, norm_layer: Optional[Callable], activation_layer: Optional[Callable],
                 group_width: int, bottleneck_multiplier: float, se_ratio: Optional[float], *, key) -> None:
        k_a, k_b, k_se, k_c = jr.split(key, 4)
        w_b = int(round(width_out * bottleneck_multiplier))
        stack: List[Module] = [
            ConvNormActivation(width_in, w_b, kernel_size=1, stride=1, norm_layer=norm_layer, activation_layer=activation_layer, key=k_a),
            ConvNormActivation(w_b, w_b, kernel_size=3, stride=stride, groups=w_b // group_width, norm_layer=norm_layer,
                               activation_layer=activation_layer, key=k_b)]
        if se_ratio:           # the SE reduction is relative to the block INPUT width (reference :74-77)
            stack.append(SqueezeExcitation(input_channels=w_b, squeeze_channels=int(round(se_ratio * width_in)),
                                           activation=activation_layer, key=k_se))
        stack.append(ConvNormActivation(w_b, width_out, kernel_size=1, stride=1, norm_layer=norm_layer, activation_layer=None, key=k_c))
        super().__init__(stack)


class ResBottleneckBlock(Module):
    proj: Module
    f: Module
    activation: Callable

    def __init__(self, width_in: int, width_out: int, stride: int, norm_layer: Optional[Callable], activation_layer: Optional[Callable],
                 group_width: int = 1, bottleneck_multiplier: float = 1.0, se_ratio: Optional[float] = None, *, key=None) -> None:
        k_proj, k_f = jr.split(key if key is not None else jr.PRNGKey(0), 2)
        self.proj = nn.Identity()
        if width_in != width_out or stride != 1:
            self.proj = ConvNormActivation(width_in, width_out, kernel_size=1, stride=stride, norm_layer=norm_layer,
                                           activation_layer=None, key=k_proj)
        self.f = BottleneckTransform(width_in, width_out, stride, norm_layer, activation_layer, group_width, bottleneck_multiplier,
                                     se_ratio, key=k_f)
        self.activation = activation_layer

    @boundary
    def __call__(self, x, *, key=None):
        x = ops.as_map(x)
        shortcut = self.proj(x)
        a = nn.act_name(self.activation)
        last = self.f.layers[-1]
        L = getattr(last, "layers", None)
        if a in ("relu", "gelu") and L is not None and len(L) == 2 and type(L[0]) is nn.Conv2d and isinstance(L[1], nn.BatchNorm) \
                and L[1].inference:
            h = self.f[:-1](x)
            return ops.conv2d(h, L[0], L[1], a, residual=shortcut)      # act(proj(x) + f(x)) in the last GEMM's epilogue
        return self.activation(ops.add(shortcut, self.f(x)))


class AnyStage(nn.Sequential):
    def __init__(self, width_in: int, width_out: int, stride: int, depth: int, block_constructor: Callable, norm_layer: Callable,
                 activation_layer: Callable, group_width: int, bottleneck_multiplier: float, se_ratio: Optional[float] = None, *,
                 key=None) -> None:
        keys = jr.split(key if key is not None else jr.PRNGKey(0), depth)
        super().__init__([block_constructor(width_in if i == 0 else width_out, width_out, stride if i == 0 else 1, norm_layer,
                                            activation_layer, group_width, bottleneck_multiplier, se_ratio, key=keys[i])
                          for i in range(depth)])


class BlockParams:
    def __init__(self, depths: List[int], widths: List[int], group_widths: List[int], bottleneck_multipliers: List[float],
                 strides: List[int], se_ratio: Optional[float] = None) -> None:
        self.depths, self.widths, self.group_widths = depths, widths, group_widths
        self.bottleneck_multipliers, self.strides, self.se_ratio = bottleneck_multipliers, strides, se_ratio

    @classmethod
    def from_init_params(cls, depth: int, w_0: int, w_a: float, w_m: float, group_width: int, bottleneck_multiplier: float = 1.0,
                         se_ratio: Optional[float] = None) -> "BlockParams":
        """The RegNet width rule (reference :197-262): block j has the continuous width w_0 + w_a * j, snapped to the nearest
        w_0 * w_m^k and then to a multiple of 8; equal consecutive widths form a stage.  float32 arithmetic like the reference's
        jnp ops (and torchvision's torch ops): the published widths sit on rounding boundaries."""
        QUANT, STRIDE = 8, 2
        if w_a < 0 or w_0 <= 0 or w_m <= 1 or w_0 % 8 != 0:
            raise ValueError("Invalid RegNet settings")
        f32 = np.float32
        cont = np.arange(depth, dtype=f32) * f32(w_a) + f32(w_0)
        capacity = np.round(np.log(cont / f32(w_0)) / np.log(f32(w_m))).astype(f32)
        block_widths = (np.round(f32(w_0) * np.power(f32(w_m), capacity) / f32(QUANT)) * QUANT).astype(np.int32).tolist()
        num_stages = len(set(block_widths))
        pairs = zip(block_widths + [0], [0] + block_widths)
        splits = [w != wp for w, wp in pairs]
        stage_widths = [w for w, t in zip(block_widths, splits[:-1]) if t]
        edges = [d for d, t in enumerate(splits) if t]
        stage_depths = [b - a for a, b in zip(edges[:-1], edges[1:])]
        multipliers = [bottleneck_multiplier] * num_stages
        stage_widths, group_widths = cls._adjust_widths_groups_compatibilty(stage_widths, multipliers, [group_width] * num_stages)
        return cls(depths=stage_depths, widths=stage_widths, group_widths=group_widths, bottleneck_multipliers=multipliers,
                   strides=[STRIDE] * num_stages, se_ratio=se_ratio)

    def _get_expanded_params(self):
        return zip(self.widths, self.strides, self.depths, self.group_widths, self.bottleneck_multipliers)

    @staticmethod
    def _adjust_widths_groups_compatibilty(stage_widths: List[int], bottleneck_ratios: List[float],
                                           group_widths: List[int]) -> Tuple[List[int], List[int]]:
        widths = [int(w * b) for w, b in zip(stage_widths, bottleneck_ratios)]
        gmin = [min(g, w) for g, w in zip(group_widths, widths)]
        fitted = [_make_divisible(w, g) for w, g in zip(widths, gmin)]
        return [int(w / b) for w, b in zip(fitted, bottleneck_ratios)], gmin


class RegNet(Module):
    stem: Module
    trunk_output: nn.Sequential
    avgpool: nn.AdaptiveAvgPool2d
    fc: Module

    def __init__(self, block_params: BlockParams, num_classes: int = 1000, stem_width: int = 32, stem_type: Optional[Callable] = None,
                 block_type: Optional[Callable] = None, norm_layer: Optional[Callable] = None, activation: Optional[Callable] = None, *,
                 key=None) -> None:
        stem_type = stem_type or SimpleStemIN
        norm_layer = norm_layer or nn.BatchNorm
        block_type = block_type or ResBottleneckBlock
        activation = activation or nn.relu
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, 2)
        self.stem = stem_type(3, stem_width, norm_layer, activation, key=keys[0])
        width, stages = stem_width, []
        for width_out, stride, depth, group_width, multiplier in block_params._get_expanded_params():
            keys = jr.split(keys[1], 2)
            stages.append(AnyStage(width, width_out, stride, depth, block_type, norm_layer, activation, group_width, multiplier,
                                   block_params.se_ratio, key=keys[0]))
            width = width_out
        self.trunk_output = nn.Sequential(stages)
        self.avgpool = nn.AdaptiveAvgPool2d((1, 1))
        self.fc = nn.Linear(in_features=width, out_features=num_classes, key=keys[1])

    def __call__(self, x, *, key):
        if key is None:                                  # the reference splits the key first thing (:392)
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x)

    @boundary
    def _forward(self, x):
        from ..._act import head_fp32
        x = self.trunk_output(self.stem(x))
        if type(self.avgpool) is nn.AdaptiveAvgPool2d and head_fp32():
            x = ops.adaptive_avgpool2d(x, self.avgpool.target_shape, out_fp32=True)
        else:
            x = self.avgpool(x)
        x = ops.flatten(x)
        return ops.linear_head(x, self.fc) if type(self.fc) is nn.Linear else self.fc(x)


def _regnet(arch: str, block_params: BlockParams, torch_weights: str, **kwargs: Any) -> RegNet:
    norm_layer = kwargs.pop("norm_layer", partial(nn.BatchNorm, eps=1e-05, momentum=0.1))
    model = RegNet(block_params, norm_layer=norm_layer, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


# name -> from_init_params arguments (reference :413-676)
_CONFIGS = {
    "regnet_y_400mf": dict(depth=16, w_0=48, w_a=27.89, w_m=2.09, group_width=8, se_ratio=0.25),
    "regnet_y_800mf": dict(depth=14, w_0=56, w_a=38.84, w_m=2.4, group_width=16, se_ratio=0.25),
    "regnet_y_1_6gf": dict(depth=27, w_0=48, w_a=20.71, w_m=2.65, group_width=24, se_ratio=0.25),
    "regnet_y_3_2gf": dict(depth=21, w_0=80, w_a=42.63, w_m=2.66, group_width=24, se_ratio=0.25),
    "regnet_y_8gf": dict(depth=17, w_0=192, w_a=76.82, w_m=2.19, group_width=56, se_ratio=0.25),
    "regnet_y_16gf": dict(depth=18, w_0=200, w_a=106.23, w_m=2.48, group_width=112, se_ratio=0.25),
    "regnet_y_32gf": dict(depth=20, w_0=232, w_a=115.89, w_m=2.53, group_width=232, se_ratio=0.25),
    "regnet_y_128gf": dict(depth=27, w_0=456, w_a=160.83, w_m=2.52, group_width=264, se_ratio=0.25),
    "regnet_x_400mf": dict(depth=22, w_0=24, w_a=24.48, w_m=2.54, group_width=16),
    "regnet_x_800mf": dict(depth=16, w_0=56, w_a=35.73, w_m=2.28, group_width=16),
    "regnet_x_1_6gf": dict(depth=18, w_0=80, w_a=34.01, w_m=2.25, group_width=24),
    "regnet_x_3_2gf": dict(depth=25, w_0=88, w_a=26.31, w_m=2.25, group_width=48),
    "regnet_x_8gf": dict(depth=23, w_0=80, w_a=49.56, w_m=2.88, group_width=120),
    "regnet_x_16gf": dict(depth=22, w_0=216, w_a=55.59, w_m=2.1, group_width=128),
    "regnet_x_32gf": dict(depth=23, w_0=320, w_a=69.86, w_m=2.0, group_width=168),
}


def _variant(name: str):
    def make(torch_weights: str = None, **kwargs: Any) -> RegNet:
        return _regnet(name, BlockParams.from_init_params(**_CONFIGS[name]), torch_weights, **kwargs)

    make.__name__ = make.__qualname__ = name
    make.__doc__ = f"{name} (reference models/classification/regnet.py); `torch_weights`: torchvision checkpoint path / URL."
    return make


globals().update({n: _variant(n) for n in _CONFIGS})
__all__ = ["RegNet", "BlockParams", "SimpleStemIN", "BottleneckTransform", "ResBottleneckBlock", "AnyStage"] + list(_CONFIGS)
