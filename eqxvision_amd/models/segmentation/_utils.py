"""Backbone + dense head + optional auxiliary head (reference models/segmentation/_utils.py:10-60).

`backbone(x)` is an `intermediate_layer_getter` wrapper: it returns `(ignored, [feature maps])`; the classifier runs on the
LAST one, the auxiliary classifier on the FIRST, and both results are resized to the input resolution with
`jax.image.resize(..., "bilinear")` -- on the device ONE kernel that reads the NHWC logits and writes the fp32 NCHW result the
caller receives (no intermediate tensor at the input resolution)."""
from __future__ import annotations

from typing import Optional

from ... import ops
from ... import random as jr
from ..._module import Module
from ...nn import boundary


class _SimpleSegmentationModel(Module):
    backbone: Module
    classifier: Module
    aux_classifier: Module

    def __init__(self, backbone: Module, classifier: Module, aux_classifier: Optional[Module] = None) -> None:
        self.backbone = backbone
        self.classifier = classifier
        self.aux_classifier = aux_classifier

    def __call__(self, x, *, key):
        if key is None:                          # the reference splits the key first thing (_utils.py:47): no key, no call
            raise RuntimeError("The model requires a PRNGKey.")
        return self._forward(x, key)

    @boundary
    def _forward(self, x, key):
        k_backbone = jr.split(jr.PRNGKey(0), 3)[0] if key is None else key
        size = tuple(x.shape[-2:])               # logical (C, H, W) of the sample
        _, feats = self.backbone(x, key=k_backbone)
        out = ops.resize_bilinear(self.classifier(feats[-1]), size, final=True)
        if self.aux_classifier is None:
            return None, out
        aux = ops.resize_bilinear(self.aux_classifier(feats[0]), size, final=True)
        return aux, out
