"""Stochastic depth (reference layers/drop_path.py:8-61).  Inference / p == 0 is the identity; the training branch draws
`bernoulli(key, 1 - p)` per sample ("global") or per entry of the sample's first axis ("local") from JAX's bit stream
(eqxvision_amd/random.py) and scales on the device (ops.drop_path)."""
from __future__ import annotations

from .._module import Module


class DropPath(Module):
    p: float
    inference: bool
    mode: str

    def __init__(self, p: float = 0.0, inference: bool = False, mode="global"):
        self.p = float(p)          # a python float on purpose: SURVEY Appendix C-4
        self.inference = inference
        self.mode = mode

    def __call__(self, x, *, key=None):
        if self.inference or self.p == 0.0:                  # reference :44-45
            return x
        if key is None:                                      # reference :46-49
            raise RuntimeError(
                "DropPath requires a key when running in non-deterministic mode. Did you mean to enable inference?")
        from .. import ops
        from .._act import is_act, wrap
        from ..nn import _unwrap
        if is_act(x):
            return ops.drop_path(x, self.p, self.mode, key)
        return _unwrap(ops.drop_path(wrap(x, False), self.p, self.mode, key), False)
