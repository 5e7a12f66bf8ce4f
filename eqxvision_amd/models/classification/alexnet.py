"""AlexNet (reference models/classification/alexnet.py:14-103): same fields / constructor /
`__call__(x, *, key)`; the forward is 5 fused conv+bias+relu implicit-GEMM launches, 3 max-pools
and 3 fused Linear(+relu) GEMMs."""
from __future__ import annotations

from typing import Any, Optional

from ... import nn, ops
from ... import random as jr
from ..._module import Module
from ...nn import boundary
from ...utils import load_torch_weights


class AlexNet(Module):
    features: Module
    avgpool: Module
    classifier: Module

    def __init__(self, num_classes: int = 1000, dropout: float = 0.5, *, key=None) -> None:
        if key is None:
            key = jr.PRNGKey(0)
        k = jr.split(key, 8)
        self.features = nn.Sequential([
            nn.Conv2d(3, 64, kernel_size=11, stride=4, padding=2, key=k[0]),
            nn.Lambda(nn.relu),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(64, 192, kernel_size=5, padding=2, key=k[1]),
            nn.Lambda(nn.relu),
            nn.MaxPool2d(kernel_size=3, stride=2),
            nn.Conv2d(192, 384, kernel_size=3, padding=1, key=k[2]),
            nn.Lambda(nn.relu),
            nn.Conv2d(384, 256, kernel_size=3, padding=1, key=k[3]),
            nn.Lambda(nn.relu),
            nn.Conv2d(256, 256, kernel_size=3, padding=1, key=k[4]),
            nn.Lambda(nn.relu),
            nn.MaxPool2d(kernel_size=3, stride=2),
        ])
        self.avgpool = nn.AdaptiveAvgPool2d((6, 6))
        self.classifier = nn.Sequential([
            nn.Dropout(p=dropout),
            nn.Linear(256 * 6 * 6, 4096, key=k[5]),
            nn.Lambda(nn.relu),
            nn.Dropout(p=dropout),
            nn.Linear(4096, 4096, key=k[6]),
            nn.Lambda(nn.relu),
            nn.Linear(4096, num_classes, key=k[7]),
        ])

    def __call__(self, x, *, key):
        if key is None:                                  # reference :78-79
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x)

    @boundary
    def _forward(self, x):
        x = self.features(x)
        x = self.avgpool(x)
        x = ops.flatten(x)                               # jnp.ravel in CHW order (reference :83)
        head = self.classifier.layers[-1]
        if type(head) is nn.Linear:                      # keep the logits in fp32
            x = self.classifier[:-1](x)
            return ops.linear(x, head, out_fp32=True)
        return self.classifier(x)

    def __iter__(self):
        for attr in self.__fields__:
            yield attr, getattr(self, attr)


def alexnet(torch_weights: str = None, **kwargs: Any) -> AlexNet:
    """`torch_weights`: path or URL of a torchvision checkpoint (reference :92-103)."""
    model = AlexNet(**kwargs)
    if torch_weights:
        model = load_torch_weights(model, torch_weights=torch_weights)
    return model
