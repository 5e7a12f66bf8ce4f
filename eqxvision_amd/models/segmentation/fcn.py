"""FCN (reference models/segmentation/fcn.py:14-120): a classification backbone made fully convolutional by dilation, an
`FCNHead` on its last stage and optionally one on the stage before.  Same constructor arguments, same validation errors."""
from __future__ import annotations

from typing import Callable, Optional

from ... import nn
from ... import random as jr
from ..._module import Module, tree_at
from ...experimental import intermediate_layer_getter
from ...utils import load_torch_weights
from ..classification import resnet
from ._utils import _SimpleSegmentationModel


class FCN(_SimpleSegmentationModel):
    pass


class FCNHead(nn.Sequential):
    """conv3x3 (C -> C/4, no bias) + BatchNorm + relu [one launch] + Dropout(0.1) + conv1x1 (-> classes)."""

    def __init__(self, in_channels: int, out_channels: int, *, key) -> None:
        k3, k1 = jr.split(key, 2)
        mid = in_channels // 4
        super().__init__([
            nn.Conv2d(in_channels, mid, 3, padding=1, use_bias=False, key=k3),
            nn.BatchNorm(mid, axis_name="batch"),
            nn.Lambda(nn.relu),
            nn.Dropout(0.1),
            nn.Conv2d(mid, out_channels, 1, key=k1),
        ])


def _check_layers(n_layers: int, aux_in_channels) -> None:
    if aux_in_channels is not None and n_layers != 2:
        raise ValueError("aux_in_channels requires the intermediate_layers to return exactly 2 layers "
                         "corresponding to aux and final.")
    if aux_in_channels is None and n_layers != 1:
        raise ValueError(f"With no aux_in_channels, the aux layer is disabled. Received {n_layers} "
                         "from intermediate_layers, expected number of layers is 1.")


def _prepare_backbone(backbone: Optional[Module], intermediate_layers: Callable, silence_layers: Optional[Callable]) -> Module:
    """Default dilated ResNet-50, classifier head silenced (an Identity holds no weights: the checkpoint has none for it),
    chosen stages wrapped so that their outputs are returned."""
    if backbone is None:
        backbone = resnet.resnet50(replace_stride_with_dilation=[False, True, True])
    if silence_layers is None:
        silence_layers = lambda m: m.fc
    backbone = tree_at(silence_layers, backbone, replace_fn=lambda _: nn.Identity())
    return intermediate_layer_getter(backbone, intermediate_layers)


def fcn(num_classes: Optional[int] = 21, backbone: Module = None, intermediate_layers: Callable = None,
        classifier_module: Module = None, classifier_in_channels: int = 2048, aux_in_channels: int = None,
        silence_layers: Callable = None, torch_weights: str = None, *, key=None) -> FCN:
    """Sample call (reference docstring): `fcn(backbone=resnet50(replace_stride_with_dilation=[False, True, True]),
    intermediate_layers=lambda x: [x.layer3, x.layer4], aux_in_channels=1024, torch_weights=...)`."""
    if key is None:
        key = jr.PRNGKey(0)
    k_main, k_aux = jr.split(key, 2)
    head = classifier_module if classifier_module is not None else FCNHead
    probe = backbone if backbone is not None else resnet.resnet50(replace_stride_with_dilation=[False, True, True])
    _check_layers(len(intermediate_layers(probe)), aux_in_channels)
    wrapped = _prepare_backbone(probe, intermediate_layers, silence_layers)
    classifier = head(in_channels=classifier_in_channels, out_channels=num_classes, key=k_main)
    aux = head(in_channels=aux_in_channels, out_channels=num_classes, key=k_aux) if aux_in_channels is not None else None
    model = FCN(wrapped, classifier, aux)
    if torch_weights:
        return load_torch_weights(model, torch_weights=torch_weights)
    return model
