"""MobileNetV2 (reference models/classification/mobilenetv2.py:16-244): same fields (`features`,
`classifier`, `pool`), constructor arguments and `__call__(x, *, key)`.  The reference's blocks use plain `relu` (not relu6,
mobilenetv2.py:54,66) -- kept.

Device lowering of an inverted-residual block: 1x1 expansion (+BN+relu) = one MFMA GEMM launch (channel counts that are not
multiples of 64 run with a zero-filled last k-tile), 3x3 depthwise (+BN+relu) = one HBM-bound vector launch (no reduction over
channels: nothing for the matrix cores), 1x1 linear projection (+BN, + the block input when `use_res_connect`) = one GEMM launch
with the residual added in its epilogue."""
from __future__ import annotations

from typing import Any, Callable, List, Optional

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...layers import ConvNormActivation
from ...nn import boundary
from ...utils import _make_divisible, load_torch_weights


class _InvertedResidual(Module):
    stride: int
    use_res_connect: int
    conv: nn.Sequential
    out_channels: int

    def __init__(self, inp: int, oup: int, stride: int, expand_ratio: int, norm_layer: Optional[Callable] = None, *, key=None) -> None:
        if stride not in (1, 2):
            raise AssertionError(f"stride should be 1 or 2, got {stride}")          # reference :37 (assert)
        k_expand, k_dw, k_project = jr.split(key if key is not None else jr.PRNGKey(0), 3)
        if norm_layer is None:
            norm_layer = nn.BatchNorm
        hidden = int(round(inp * expand_ratio))
        self.stride = stride
        self.use_res_connect = stride == 1 and inp == oup
        stack: List[Module] = []
        if expand_ratio != 1:
            stack.append(ConvNormActivation(inp, hidden, kernel_size=1, norm_layer=norm_layer, activation_layer=nn.relu, key=k_expand))
        stack.append(ConvNormActivation(hidden, hidden, stride=stride, groups=hidden, norm_layer=norm_layer,
                                        activation_layer=nn.relu, key=k_dw))
        stack.append(nn.Conv2d(hidden, oup, 1, 1, 0, use_bias=False, key=k_project))
        stack.append(norm_layer(oup, axis_name="batch"))
        self.conv = nn.Sequential(stack)
        self.out_channels = oup

    @boundary
    def __call__(self, x, *, key=None):
        if not self.use_res_connect:
            return self.conv(x, key=key)
        L = self.conv.layers
        if type(L[-2]) is nn.Conv2d and isinstance(L[-1], nn.BatchNorm) and L[-1].inference:
            x = ops.as_map(x)
            h = self.conv[:-2](x, key=key)
            return ops.conv2d(h, L[-2], L[-1], None, residual=x)       # x + bn(project(h)): the add rides in the GEMM epilogue
        return ops.add(x, self.conv(x, key=key))


class MobileNetV2(Module):
    features: nn.Sequential
    classifier: nn.Sequential
    pool: nn.AdaptiveAvgPool2d

    def __init__(self, num_classes: int = 1000, width_mult: float = 1.0, inverted_residual_setting: Optional[List[List[int]]] = None,
                 round_nearest: int = 8, block: Optional[Callable] = None, norm_layer: Optional[Callable] = None,
                 dropout: float = 0.2, *, key=None) -> None:
        if key is None:
            key = jr.PRNGKey(0)
        keys = jr.split(key, 2)
        block = block or _InvertedResidual
        norm_layer = norm_layer or nn.BatchNorm
        if inverted_residual_setting is None:
            #                            t, c, n, s   (expansion, channels, repeats, stride of the first repeat)
            inverted_residual_setting = [[1, 16, 1, 1], [6, 24, 2, 2], [6, 32, 3, 2], [6, 64, 4, 2], [6, 96, 3, 1], [6, 160, 3, 2],
                                         [6, 320, 1, 1]]
        if len(inverted_residual_setting) == 0 or len(inverted_residual_setting[0]) != 4:
            raise ValueError(f"inverted_residual_setting should be non-empty or a 4-element list, got {inverted_residual_setting}")
        cin = _make_divisible(32 * width_mult, round_nearest)
        last = _make_divisible(1280 * max(1.0, width_mult), round_nearest)
        stack: List[Module] = [ConvNormActivation(3, cin, stride=2, norm_layer=norm_layer, activation_layer=nn.relu, key=keys[0])]
        for t, c, n, s in inverted_residual_setting:
            cout = _make_divisible(c * width_mult, round_nearest)
            for i in range(n):
                keys = jr.split(keys[1], 2)
                stack.append(block(cin, cout, s if i == 0 else 1, expand_ratio=t, norm_layer=norm_layer, key=keys[0]))
                cin = cout
        keys = jr.split(keys[1], 2)
        stack.append(ConvNormActivation(cin, last, kernel_size=1, norm_layer=norm_layer, activation_layer=nn.relu, key=keys[0]))
        self.features = nn.Sequential(stack)
        self.classifier = nn.Sequential([nn.Dropout(p=dropout), nn.Linear(last, num_classes, key=keys[1])])
        self.pool = nn.AdaptiveAvgPool2d((1, 1))

    def __call__(self, x, *, key):
        if key is None:                                  # the reference splits the key first thing (mobilenetv2.py:222)
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x, key)

    @boundary
    def _forward(self, x, key=None):
        from ..._act import head_fp32
        from ...transforms import _needs_eager
        if key is not None and _needs_eager(self):       # training mode: the classifier's Dropout draws from keys[2]
            keys = jr.split(key, 3)                      # reference mobilenetv2.py:223-227
            x = self.features(x, key=keys[0])
            return self.classifier(ops.flatten(self.pool(x)), key=keys[2])
        x = self.features(x)
        if type(self.pool) is nn.AdaptiveAvgPool2d and head_fp32():
            x = ops.adaptive_avgpool2d(x, self.pool.target_shape, out_fp32=True)
        else:
            x = self.pool(x)
        x = ops.flatten(x)
        head = self.classifier.layers[-1]
        if type(head) is nn.Linear:
            x = self.classifier[:-1](x)
            return ops.linear_head(x, head)
        return self.classifier(x)


def mobilenet_v2(torch_weights: str = None, **kwargs: Any) -> MobileNetV2:
    model = MobileNetV2(**kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model
