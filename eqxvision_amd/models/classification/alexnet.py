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
        keys = iter(jr.split(key, 8))
        # (out_channels, kernel, stride, padding, max-pool after?) -- torchvision's AlexNet; the layer ORDER is the
        # load_torch_weights contract (ordered zip with the checkpoint)
        plan = ((64, 11, 4, 2, True), (192, 5, 1, 2, True), (384, 3, 1, 1, False), (256, 3, 1, 1, False), (256, 3, 1, 1, True))
        stack, cin = [], 3
        for cout, ksz, st, pad, pool in plan:
            stack += [nn.Conv2d(cin, cout, kernel_size=ksz, stride=st, padding=pad, key=next(keys)), nn.Lambda(nn.relu)]
            if pool:
                stack.append(nn.MaxPool2d(kernel_size=3, stride=2))
            cin = cout
        self.features = nn.Sequential(stack)
        self.avgpool = nn.AdaptiveAvgPool2d((6, 6))
        head, width = [], cin * 6 * 6
        for hidden in (4096, 4096):
            head += [nn.Dropout(p=dropout), nn.Linear(width, hidden, key=next(keys)), nn.Lambda(nn.relu)]
            width = hidden
        head.append(nn.Linear(width, num_classes, key=next(keys)))
        self.classifier = nn.Sequential(head)

    def __call__(self, x, *, key):
        if key is None:                                  # reference :78-79
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x, key)

    @boundary
    def _forward(self, x, key=None):
        from ...transforms import _needs_eager
        if key is not None and _needs_eager(self):       # training mode: the classifier's Dropouts draw from keys[1] (alexnet.py:80-84)
            keys = jr.split(key, 2)
            x = self.features(x, key=keys[0])
            x = self.avgpool(x)
            return self.classifier(ops.flatten(x), key=keys[1])
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
