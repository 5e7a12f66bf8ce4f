"""VGG 11/13/16/19 with and without BatchNorm (reference models/classification/vgg.py:15-144 and the `vggNN[_bn]`
constructors below it).  Same fields (`features`, `avgpool`, `classifier`), same constructor
arguments, same `__call__(x, *, key)`.

On the device every `Conv2d [-> BatchNorm] -> relu` triple of `features` is ONE implicit-GEMM launch (the `nn.Sequential`
peephole, BatchNorm folded into the fp32 scale/shift epilogue), the first 3-channel convolution reads the NCHW fp32 image
directly (stem kernel), and the classifier is three GEMMs.

Kept from the reference, deliberately: the classifier is `Linear, Dropout, Linear, relu, Dropout, Linear` -- ONE relu, after
the second Linear (vgg.py:96-105; torchvision has a relu after the first Linear as well).  A torchvision checkpoint loads
(the ordered zip only sees the three Linear layers) but evaluates differently there, exactly as in the reference.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Union

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...nn import boundary
from ...utils import load_torch_weights

# torchvision's letter-coded layer plans: an int is a 3x3 convolution with that many output channels, "M" a 2x2 max-pool
_PLANS: Dict[str, str] = {
    "A": "64 M 128 M 256 256 M 512 512 M 512 512 M",
    "B": "64 64 M 128 128 M 256 256 M 512 512 M 512 512 M",
    "D": "64 64 M 128 128 M 256 256 256 M 512 512 512 M 512 512 512 M",
    "E": "64 64 M 128 128 M 256 256 256 256 M 512 512 512 512 M 512 512 512 512 M",
}
_cfgs: Dict[str, List[Union[str, int]]] = {k: [t if t == "M" else int(t) for t in v.split()] for k, v in _PLANS.items()}


def _make_layers(cfg: List[Union[str, int]], batch_norm: bool = False, key=None) -> nn.Sequential:
    """reference vgg.py:122-148: conv keys are drawn in layer order from one split of `key`."""
    widths = [v for v in cfg if v != "M"]
    keys = iter(jr.split(key if key is not None else jr.PRNGKey(0), max(len(widths), 1)))
    stack: List[Module] = []
    cin = 3
    for v in cfg:
        if v == "M":
            stack.append(nn.MaxPool2d(kernel_size=2, stride=2))
            continue
        stack.append(nn.Conv2d(cin, int(v), kernel_size=3, padding=1, key=next(keys)))
        if batch_norm:
            stack.append(nn.BatchNorm(int(v), axis_name="batch"))
        stack.append(nn.Lambda(nn.relu))
        cin = int(v)
    return nn.Sequential(stack)


class VGG(Module):
    features: nn.Sequential
    avgpool: nn.AdaptiveAvgPool2d
    classifier: nn.Sequential

    def __init__(self, cfg: List[Union[str, int]] = None, num_classes: int = 1000, batch_norm: bool = True,
                 dropout: float = 0.5, *, key=None) -> None:
        if cfg is None:
            raise ValueError("VGG needs a layer plan (`cfg`), e.g. one of the vggNN constructors")
        if key is None:
            key = jr.PRNGKey(0)
        k_feat, k1, k2, k3 = jr.split(key, 4)
        self.features = _make_layers(cfg, batch_norm, key=k_feat)
        self.avgpool = nn.AdaptiveAvgPool2d((7, 7))
        width = [v for v in cfg if v != "M"][-1]
        self.classifier = nn.Sequential([
            nn.Linear(int(width) * 7 * 7, 4096, key=k1),
            nn.Dropout(p=dropout),
            nn.Linear(4096, 4096, key=k2),
            nn.Lambda(nn.relu),
            nn.Dropout(p=dropout),
            nn.Linear(4096, num_classes, key=k3),
        ])

    def __call__(self, x, *, key):
        # the reference splits `key` unconditionally (vgg.py:113): a missing key is an error there too
        if key is None:
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x, key)

    @boundary
    def _forward(self, x, key=None):
        from ...transforms import _needs_eager
        if key is not None and _needs_eager(self):       # training mode: the classifier's Dropouts draw from keys[1] (vgg.py:113-118)
            keys = jr.split(key, 2)
            x = self.features(x, key=keys[0])
            x = self.avgpool(x)
            return self.classifier(ops.flatten(x), key=keys[1])
        x = self.features(x)
        x = self.avgpool(x)
        x = ops.flatten(x)                               # jnp.ravel in CHW order (vgg.py:116)
        head = self.classifier.layers[-1]
        if type(head) is nn.Linear:                      # keep the logits in fp32
            x = self.classifier[:-1](x)
            return ops.linear(x, head, out_fp32=True)
        return self.classifier(x)


def _vgg(cfg: str, batch_norm: bool, torch_weights: Optional[str], **kwargs: Any) -> VGG:
    model = VGG(cfg=_cfgs[cfg], batch_norm=batch_norm, **kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model


def vgg11(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("A", False, torch_weights, **kwargs)


def vgg11_bn(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("A", True, torch_weights, **kwargs)


def vgg13(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("B", False, torch_weights, **kwargs)


def vgg13_bn(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("B", True, torch_weights, **kwargs)


def vgg16(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("D", False, torch_weights, **kwargs)


def vgg16_bn(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("D", True, torch_weights, **kwargs)


def vgg19(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("E", False, torch_weights, **kwargs)


def vgg19_bn(torch_weights: str = None, **kwargs: Any) -> VGG:
    return _vgg("E", True, torch_weights, **kwargs)
