"""`eqxvision.experimental.intermediate_layer_getter` (reference experimental.py:8-88): wrap chosen sub-modules so that a
forward also returns their outputs.  Same contract: `get_target_layers(model)` returns the target layers (or, for an
`nn.Sequential`, their indices); the returned module's call gives `(output, [intermediate activations in order])`; only the
most recent call of a layer is kept.  On the device an intermediate is whatever the wrapped layer returns at the module
boundary -- a fp32 torch tensor in the reference's logical layout ((C,H,W) per sample, batched under `vmap`)."""
from __future__ import annotations

from typing import Any, Callable

from . import nn
from ._module import Module, tree_at
from .nn import boundary


class AuxData:
    """A simple container for auxiliary data (reference experimental.py:8-20)."""

    def __init__(self):
        self.data = None

    def update(self, x: Any):
        self.data = x


def _make_intermediate_layer_wrapper():
    aux = AuxData()

    class IntermediateWrapper(Module):
        layer: Module

        def __init__(self, layer):
            self.layer = layer

        @boundary
        def __call__(self, x, *, key=None):
            out = self.layer(x, key=key)
            aux.update(out)
            return out

    return aux, IntermediateWrapper


def intermediate_layer_getter(model: Module, get_target_layers: Callable) -> Module:
    target_layers = get_target_layers(model)
    auxs, wrappers = zip(*[_make_intermediate_layer_wrapper() for _ in range(len(target_layers))])
    if isinstance(model, nn.Sequential):
        new_modules, updated = [], 0
        for idx, module in enumerate(model.layers):
            if idx in target_layers:
                new_modules.append(wrappers[updated](module))
                updated += 1
            else:
                new_modules.append(module)
        model = nn.Sequential(new_modules)
    else:
        model = tree_at(get_target_layers, model, [w(t) for w, t in zip(wrappers, target_layers)])

    class IntermediateLayerGetter(Module):
        model: Module

        def __init__(self, model):
            self.model = model

        def __call__(self, x, *, key=None):
            out = self.model(x, key=key)
            return out, [aux.data for aux in auxs]

    return IntermediateLayerGetter(model)
