"""Stochastic depth (reference layers/drop_path.py:8-61).  Inference / p == 0 is the identity,
which is all the forward hot path needs; the Bernoulli training branch is out of scope."""
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
        raise NotImplementedError(
            "DropPath's random training branch is outside the inference hot path; "
            "use eqxvision_amd.tree_inference(model, True)")
