"""Transformer MLP (reference layers/mlps.py:12-66): fc1 -> act -> drop -> fc2 -> drop.

On the device this is two GEMM launches: bias + activation are fused into fc1's epilogue, bias
(+ the caller's residual, see `_forward`) into fc2's."""
from __future__ import annotations

from typing import Callable, Tuple, Union


from .. import nn, ops
from .. import random as jr
from .._module import Module
from ..nn import boundary


class MlpProjection(Module):
    fc1: Module
    act: Callable
    drop1: nn.Dropout
    fc2: Module
    drop2: nn.Dropout

    def __init__(self, in_features: int, hidden_features: int = None, out_features: int = None,
                 lin_layer=nn.Linear, act_layer: Callable = None, drop: Union[float, Tuple[float]] = 0.0,
                 *, key=None):
        out_features = out_features or in_features
        hidden_features = hidden_features or in_features
        drop_probs = drop if isinstance(drop, tuple) else (drop, drop)
        keys = jr.split(key if key is not None else jr.PRNGKey(0), 2)
        self.fc1 = lin_layer(in_features, hidden_features, key=keys[0])
        self.act = act_layer
        self.drop1 = nn.Dropout(drop_probs[0])
        self.fc2 = lin_layer(hidden_features, out_features, key=keys[1])
        self.drop2 = nn.Dropout(drop_probs[1])

    def _live(self) -> bool:
        return nn.dropout_live(self.drop1) or nn.dropout_live(self.drop2)

    def _forward(self, x, residual=None, norm=None, keys=None, per_row=False, precise=False):
        """`norm` given: x is the un-normalised input; the LayerNorm is folded into fc1 where the library can.
        `keys` (training mode with a live Dropout): one PRNG key per sample, or per ROW of a (tokens, features) input when
        `per_row` -- the reference's ViT vmaps the layer over the tokens (vit.py:155); each is split in two for drop1 / drop2
        (mlps.py:60-65)."""
        if x.kind in ("img", "map"):
            x = ops.as_map(x)
        name = nn.act_name(self.act)
        if norm is not None and not isinstance(self.fc1, nn.Linear):
            x, norm = norm(x), None
        if precise and name is not None and not self._live():          # split-precision weights (ops.swin_precise: swin_b's widths)
            h = ops.linear_split(x if norm is None else ops.layernorm(x, norm), self.fc1, act=name)
            return ops.linear_split(h, self.fc2, residual=residual)
        if name is not None:
            h = ops.linear(x, self.fc1, act=name) if norm is None else ops.ln_linear(x, norm, self.fc1, act=name)
        else:
            h = ops.linear(x, self.fc1) if norm is None else ops.ln_linear(x, norm, self.fc1)
            h = self.act(h) if self.act is not None else h
        if not self._live():                                           # identity in inference / p = 0 (mlps.py:63, :65)
            return ops.linear(h, self.fc2, residual=residual)
        if keys is None:
            raise RuntimeError("Dropout requires a key when running in non-deterministic mode.")
        # A map reaches an MLP through Linear2d layers (Swin, swin.py:562-570).  Linear2d hands back `(out_features, h, w)`
        # (extensions_2d.py:46-50), so eqx.nn.Dropout draws its mask for a (C, H, W) array: word c * H * W + hw of the sample's
        # stream decides element (c, hw) -- the same logical index every other feature map uses (round-2 advice: the (H, W, C)
        # order used here before dropped different elements than the reference for the same key).
        if h.kind not in ("seq", "vec") and not (h.kind == "map" and type(self.fc1).__name__ == "Linear2d"):
            raise NotImplementedError(f"MlpProjection: training-mode Dropout on a {h.kind} input is not built")
        ks = ops.split_keys(keys, 2)
        per_row = per_row and h.kind == "seq"
        if nn.dropout_live(self.drop1):
            h = ops.dropout(h, self.drop1.p, ks[0], per_row=per_row)
        y = ops.linear(h, self.fc2)
        if nn.dropout_live(self.drop2):
            y = ops.dropout(y, self.drop2.p, ks[1], per_row=per_row)
        return y if residual is None else ops.add(residual, y)

    @boundary
    def __call__(self, x, *, key=None):
        return self._forward(x, keys=key)
