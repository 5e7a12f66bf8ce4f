"""`eqxvision.layers` surface for the hot path: the reference's public names (eqxvision/layers/__init__.py), each bound
to this package's implementation module."""
from . import conv_norm_activation as _cna
from . import drop_path as _dp
from . import extensions_2d as _e2d
from . import mlps as _mlp
from . import patch_embed as _pe
from . import squeeze as _se

_EXPORTS = {
    "ConvNormActivation": _cna, "DropPath": _dp, "LayerNorm2d": _e2d, "Linear2d": _e2d, "MlpProjection": _mlp,
    "PatchEmbed": _pe, "SqueezeExcitation": _se,
}
globals().update({name: getattr(mod, name) for name, mod in _EXPORTS.items()})
__all__ = sorted(_EXPORTS)
