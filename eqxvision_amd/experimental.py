"""`eqxvision.experimental.intermediate_layer_getter` (reference experimental.py:35-88) -- public contract only:

    getter = intermediate_layer_getter(model, get_target_layers)
    out, feats = getter(x, key=key)          # feats: outputs of the target layers, in the order get_target_layers listed them
                                             # (for an `nn.Sequential` model: in LAYER order, as the reference hands its wrappers out)

`get_target_layers(model)` returns the target sub-modules (or, for an `nn.Sequential`, their indices); a layer that runs several
times in one forward reports its LAST output.

Own design (not the reference's per-layer closure classes): ONE `_Capture` frame per getter, shared by every copy of the tree
(`tree_inference` shares non-field attributes), and ONE module type, `_Tap`, that forwards to the layer it stands in for and
drops the result into its numbered slot of the frame.  A call of the getter opens a fresh frame, runs the model, and hands the
slots back -- so nothing leaks from one call into the next and a layer that did not run reports `None` instead of a stale tensor.
On the device a captured value is whatever the layer returns at the module boundary: under `vmap` a fp32 tensor in the
reference's logical layout, (B, C, H, W) for a feature map."""
from __future__ import annotations

from typing import Callable, List, Sequence

from . import nn
from ._module import Module, tree_at
from .nn import boundary


class _Capture:
    """Slots of the forward in flight.  Deliberately NOT a Module field: module copies share it by reference."""

    __slots__ = ("slots",)

    def __init__(self, n: int):
        self.slots: List = [None] * n

    def open(self):
        self.slots = [None] * len(self.slots)

    def take(self) -> list:
        return list(self.slots)


class _Tap(Module):
    """Stands where a target layer stood: same call signature, same result, plus a copy of the result in slot `slot`."""
    inner: Module
    slot: int

    def __init__(self, inner: Module, slot: int, frame: _Capture):
        self.inner = inner
        self.slot = slot
        object.__setattr__(self, "_frame", frame)

    @boundary
    def __call__(self, x, *, key=None):
        y = self.inner(x, key=key)
        self._frame.slots[self.slot] = y
        return y


class LayerGetter(Module):
    model: Module
    n_taps: int

    def __init__(self, model: Module, frame: _Capture):
        self.model = model
        self.n_taps = len(frame.slots)
        object.__setattr__(self, "_frame", frame)

    def __call__(self, x, *, key=None):
        self._frame.open()
        out = self.model(x, key=key)
        return out, self._frame.take()

    def aux(self) -> List[AuxData]:
        """The last call's captures as the reference's `AuxData` holders."""
        out = []
        for v in self._frame.take():
            a = AuxData()
            a.update(v)
            out.append(a)
        return out


class AuxData:
    """Public name kept for API parity with the reference (experimental.py:8-21: one mutable `.data` holder per target layer).
    The getter here keeps its captures in one `_Capture` frame; this holder is what `LayerGetter.aux()` hands out."""

    def __init__(self):
        self.data = None

    def update(self, x):
        self.data = x


def _tap_sequential(seq: nn.Sequential, indices: Sequence[int], frame: _Capture) -> nn.Sequential:
    # the reference walks the layers and gives the next unused wrapper to every layer whose index is a target
    # (experimental.py:60-68): slots follow LAYER order whatever order the indices were listed in, and a repeated index
    # leaves the surplus slots at the end empty (None)
    order = {i: k for k, i in enumerate(sorted({int(i) for i in indices}))}
    return nn.Sequential([_Tap(layer, order[i], frame) if i in order else layer for i, layer in enumerate(seq.layers)])


def intermediate_layer_getter(model: Module, get_target_layers: Callable) -> Module:
    targets = list(get_target_layers(model))
    frame = _Capture(len(targets))
    if isinstance(model, nn.Sequential):
        return LayerGetter(_tap_sequential(model, targets, frame), frame)
    taps = [_Tap(layer, k, frame) for k, layer in enumerate(targets)]
    return LayerGetter(tree_at(get_target_layers, model, taps), frame)
