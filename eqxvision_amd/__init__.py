"""eqxvision_amd -- MI355X-native forward-pass engine behind eqxvision's model surface.

Drop-in for the reference's vmap'd inference path (README.md:37-40 of paganpasta/eqxvision):

    import eqxvision_amd as eqv
    net = eqv.tree_inference(eqv.models.resnet50(torch_weights=...), True)
    logits = eqv.filter_jit(lambda net, x, k: eqv.vmap(net, axis_name="batch")(x, key=k))(net, images, keys)

Everything below the module `__call__`s is hand-written HIP for gfx950 behind a C ABI
(`include/eqxvision_amd.h`); there is no CPU fallback.
"""
__version__ = "0.1.0"

from . import layers, models, nn, random, utils
from ._act import (compute_dtype, precision, set_compute_dtype, set_head_fp32, set_residual_fp32, set_split_weights)
from ._module import Module, tree_at, tree_inference, tree_leaves
from .transforms import filter_jit, vmap
from . import grad as optim          # optax-shaped names: optim.adam, optim.softmax_cross_entropy, optim.one_hot
from .grad import apply_updates, filter_value_and_grad
from ._module import is_array


def filter(tree, predicate=None):
    """`eqx.filter(tree, eqx.is_array)` as the reference's training loop uses it (tests/test_grads.py:53): the optimiser here
    skips non-array leaves itself, so the tree is returned as it is."""
    return tree

__all__ = ["layers", "models", "nn", "random", "utils", "Module", "tree_at", "tree_inference", "tree_leaves",
           "filter_jit", "vmap", "filter_value_and_grad", "apply_updates", "optim", "filter", "is_array", "compute_dtype", "precision", "set_compute_dtype", "set_head_fp32", "set_residual_fp32",
           "set_split_weights"]
