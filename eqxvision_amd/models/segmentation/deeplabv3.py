"""DeepLabV3 (reference models/segmentation/deeplabv3.py:18-227): dilated backbone + ASPP head (+ FCN auxiliary head).

Device lowering of the ASPP (reference :85-136): five branches on the same 2048-channel map -- a 1x1 conv, three dilated 3x3
convs (rates 12 / 24 / 36, padding = rate), and global-average-pool -> 1x1 conv -> broadcast back -- each one fused
conv+BN+relu launch; their outputs are placed side by side in one NHWC buffer (`jnp.concatenate` along channels) and
projected by a 1x1 conv+BN+relu."""
from __future__ import annotations

from typing import Callable, List, Optional

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...nn import boundary
from ...utils import load_torch_weights
from ._utils import _SimpleSegmentationModel
from .fcn import FCNHead, _check_layers, _prepare_backbone


class DeepLabV3(_SimpleSegmentationModel):
    pass


def _conv_bn_relu(cin: int, cout: int, ksize: int, *, dilation: int = 1, key) -> List[Module]:
    pad = dilation if ksize == 3 else 0
    return [nn.Conv2d(cin, cout, ksize, padding=pad, dilation=dilation, use_bias=False, key=key),
            nn.BatchNorm(cout, axis_name="batch"), nn.Lambda(nn.relu)]


class ASPPConv(nn.Sequential):
    def __init__(self, in_channels: int, out_channels: int, dilation: int, key=None) -> None:
        super().__init__(_conv_bn_relu(in_channels, out_channels, 3, dilation=dilation, key=key))


class ASPPPooling(nn.Sequential):
    def __init__(self, in_channels: int, out_channels: int, key=None) -> None:
        super().__init__([nn.AdaptiveAvgPool2d(1)] + _conv_bn_relu(in_channels, out_channels, 1, key=key))

    @boundary
    def __call__(self, x, *, key=None):
        size = tuple(x.shape[-2:])
        y = nn.Sequential.call_chained(self, x, None)
        return ops.resize_bilinear(y, size)                       # 1x1 -> HxW: every pixel is the pooled value


class ASPP(Module):
    convs: Module
    project: Module

    def __init__(self, in_channels: int, atrous_rates: List[int], out_channels: int = 256, key=None) -> None:
        if key is None:
            key = jr.PRNGKey(0)
        rates = tuple(atrous_rates)
        keys = jr.split(key, len(rates) + 3)
        branches: List[Module] = [nn.Sequential(_conv_bn_relu(in_channels, out_channels, 1, key=keys[0]))]
        for i, rate in enumerate(rates):
            branches.append(ASPPConv(in_channels, out_channels, rate, key=keys[i + 1]))
        branches.append(ASPPPooling(in_channels, out_channels, key=keys[-2]))
        self.convs = nn.Sequential(branches)
        self.project = nn.Sequential(_conv_bn_relu(len(branches) * out_channels, out_channels, 1, key=keys[-1])
                                     + [nn.Dropout(0.5)])

    @boundary
    def __call__(self, x, *, key=None):
        x = ops.as_map(x)
        return self.project(ops.concat_channels([branch(x) for branch in self.convs.layers]), key=key)


class DeepLabHead(nn.Sequential):
    def __init__(self, in_channels: int, out_channels: int, key=None) -> None:
        k_aspp, k3, k1 = jr.split(key if key is not None else jr.PRNGKey(0), 3)
        super().__init__([ASPP(in_channels, [12, 24, 36], key=k_aspp)] + _conv_bn_relu(256, 256, 3, key=k3)
                         + [nn.Conv2d(256, out_channels, 1, key=k1)])


def deeplabv3(num_classes: Optional[int] = 21, backbone: Module = None, intermediate_layers: Callable = None,
              classifier_module: Module = None, classifier_in_channels: int = 2048, aux_classifier_module: Module = None,
              aux_in_channels: int = 1024, silence_layers: Callable = None, torch_weights: str = None, *, key=None) -> DeepLabV3:
    """Sample call (reference docstring): `deeplabv3(intermediate_layers=lambda x: [x.layer3, x.layer4], aux_in_channels=1024,
    torch_weights=...)`; the default backbone is the dilated ResNet-50."""
    if key is None:
        key = jr.PRNGKey(0)
    k_main, k_aux = jr.split(key, 2)
    head = classifier_module or DeepLabHead
    aux_head = aux_classifier_module or FCNHead
    from ..classification import resnet
    probe = backbone if backbone is not None else resnet.resnet50(replace_stride_with_dilation=[False, True, True])
    _check_layers(len(intermediate_layers(probe)), aux_in_channels)
    wrapped = _prepare_backbone(probe, intermediate_layers, silence_layers)
    classifier = head(in_channels=classifier_in_channels, out_channels=num_classes, key=k_main)
    aux = aux_head(in_channels=aux_in_channels, out_channels=num_classes, key=k_aux) if aux_in_channels is not None else None
    model = DeepLabV3(wrapped, classifier, aux)
    if torch_weights:
        return load_torch_weights(model, torch_weights=torch_weights)
    return model
