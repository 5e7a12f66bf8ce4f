"""CPU ORACLE (test infrastructure) for the BACKWARD half: torch.autograd on the fp32 `torch.nn.functional` restatement of
oracle/torch_ref.py.  loss = mean over the batch of softmax cross-entropy against one-hot labels, as the reference's gradient
test computes it (reference tests/test_grads.py:37-41: optax.softmax_cross_entropy(output, one_hot).mean()).  BatchNorm in inference mode uses the
stored running statistics as constants; the TRAINING branch (`bn_train`, `resnet_train`) normalises with the statistics it has just
updated from the batch and differentiates through the batch moments.  Returns (loss, {state-dict name: gradient}) for every floating-point
parameter.  `dtype`: torch.float32 (default: the same arithmetic as the HIP path, so the ReLU / max-pool decisions of the two sides
coincide almost everywhere) or torch.float64 (the exact gradient of the fp64 function; where an fp32 activation sits within
rounding of zero the two evaluations take different branches, which shows up as 1e-3..1e-2 differences in single tensors of deep
ReLU stacks -- measured on vgg11 / resnet50; it is not an error of either side).  PARITY UNPINNED -- see oracle/np_ops.py.  Never imported by the product."""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F

from . import torch_ref as TR


def _params(sd, dtype=torch.float32):
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.ascontiguousarray(np.asarray(v))).clone()
        if t.dtype.is_floating_point:
            t = t.to(dtype)
            if "running" not in k:
                t.requires_grad_(True)
        out[k] = t
    return out


def _finish(logits, labels, P):
    loss = F.cross_entropy(logits, torch.as_tensor(np.asarray(labels), dtype=torch.long), reduction="mean")
    loss.backward()
    return float(loss.detach()), {k: v.grad.numpy() for k, v in P.items() if v.requires_grad and v.grad is not None}


def alexnet(sd, x, labels, dtype=torch.float32):
    P = _params(sd, dtype)
    t = TR._t(P)
    x = torch.as_tensor(x).to(dtype)
    x = F.relu(F.conv2d(x, t["features.0.weight"], t["features.0.bias"], 4, 2))
    x = F.max_pool2d(x, 3, 2)
    x = F.relu(F.conv2d(x, t["features.3.weight"], t["features.3.bias"], 1, 2))
    x = F.max_pool2d(x, 3, 2)
    x = F.relu(F.conv2d(x, t["features.6.weight"], t["features.6.bias"], 1, 1))
    x = F.relu(F.conv2d(x, t["features.8.weight"], t["features.8.bias"], 1, 1))
    x = F.relu(F.conv2d(x, t["features.10.weight"], t["features.10.bias"], 1, 1))
    x = F.max_pool2d(x, 3, 2)
    x = F.adaptive_avg_pool2d(x, (6, 6)).flatten(1)
    x = F.relu(F.linear(x, t["classifier.1.weight"], t["classifier.1.bias"]))
    x = F.relu(F.linear(x, t["classifier.4.weight"], t["classifier.4.bias"]))
    return _finish(F.linear(x, t["classifier.6.weight"], t["classifier.6.bias"]), labels, P)


def resnet(sd, x, labels, block="basic", layers=(2, 2, 2, 2), dtype=torch.float32):
    P = _params(sd, dtype)
    t = TR._t(P)
    y = TR._resnet_stages(t, torch.as_tensor(x).to(dtype), block, layers)[-1]
    y = F.adaptive_avg_pool2d(y, 1).flatten(1)
    return _finish(F.linear(y, t["fc.weight"], t["fc.bias"]), labels, P)


def bn_train(sd, x, name, first, momentum=0.99, eps=1e-5, new_running=None):
    """eqx.experimental.BatchNorm, TRAINING branch (restated from equinox 0.9's experimental/batch_norm.py):
    per-channel batch mean and mean squared deviation over (batch, H, W); running' = batch on the layer's first call, else
    (1 - momentum) * batch + momentum * running; the layer normalises with running' -- and nothing stops the gradient through the
    batch moments, so autograd on this function carries the batch-statistics terms (in full on the first call, scaled by
    1 - momentum afterwards).  `new_running[name]` receives the updated statistics (detached)."""
    mean = x.mean((0, 2, 3))
    var = ((x - mean[None, :, None, None]) ** 2).mean((0, 2, 3))
    if first:
        rm, rv = mean, var
    else:
        rm = (1.0 - momentum) * mean + momentum * sd[name + ".running_mean"]
        rv = (1.0 - momentum) * var + momentum * sd[name + ".running_var"]
    if new_running is not None:
        new_running[name] = (rm.detach().numpy().copy(), rv.detach().numpy().copy())
    y = (x - rm[None, :, None, None]) / torch.sqrt(rv[None, :, None, None] + eps)
    return y * sd[name + ".weight"][None, :, None, None] + sd[name + ".bias"][None, :, None, None]


def resnet_train(sd, x, labels, block="basic", layers=(2, 2, 2, 2), first=True, momentum=0.99, dtype=torch.float32):
    """loss / gradients of the ResNet restatement with every BatchNorm in TRAINING mode (bn_train) -> (loss, grads, new_running)."""
    P = _params(sd, dtype)
    t = TR._t(P)
    new_running = {}
    saved = TR._bn
    TR._bn = lambda sd_, x_, name, eps=1e-5: bn_train(sd_, x_, name, first, momentum, eps, new_running)
    try:
        y = TR._resnet_stages(t, torch.as_tensor(x).to(dtype), block, layers)[-1]
    finally:
        TR._bn = saved
    y = F.adaptive_avg_pool2d(y, 1).flatten(1)
    loss, grads = _finish(F.linear(y, t["fc.weight"], t["fc.bias"]), labels, P)
    return loss, grads, new_running


def vit(sd, x, labels, patch=16, num_heads=3, depth=12, dtype=torch.float32):
    P = _params(sd, dtype)
    t = TR._t(P)
    y = F.conv2d(torch.as_tensor(x).to(dtype), t["patch_embed.proj.weight"], t["patch_embed.proj.bias"], patch)
    y = y.flatten(2).transpose(1, 2)
    D = y.shape[-1]
    y = torch.cat([t["cls_token"].reshape(1, 1, D).expand(y.shape[0], -1, -1), y], 1) + t["pos_embed"].reshape(1, -1, D)
    for i in range(depth):
        y = TR._vit_block(t, y, f"blocks.{i}", num_heads)
    y = F.layer_norm(y, (D,), t["norm.weight"], t["norm.bias"], 1e-5)[:, 0]
    return _finish(F.linear(y, t["fc.weight"], t["fc.bias"]), labels, P)


def vgg(sd, x, labels, plan="A", batch_norm=False, dtype=torch.float32):
    from .state import VGG_PLANS
    P = _params(sd, dtype)
    logits = TR.vgg_forward.__wrapped__(P, torch.as_tensor(x).to(dtype), VGG_PLANS[plan] if isinstance(plan, str) else tuple(plan), batch_norm)
    return _finish(logits, labels, P)


def mobilenet_v2(sd, x, labels, setting=None, dtype=torch.float32):
    """Depthwise / pointwise inverted residuals (mobilenetv2.py): exercises the grouped-convolution gradients."""
    from .state import MBV2_SETTING
    P = _params(sd, dtype)
    logits = TR.mobilenet_v2_forward.__wrapped__(P, torch.as_tensor(x).to(dtype), setting if setting is not None else MBV2_SETTING)
    return _finish(logits, labels, P)


def swin(sd, x, labels, dtype=torch.float32, **kw):
    """Shifted-window attention with the relative-position bias table, patch merging, LayerNorm2d (swin.py)."""
    P = _params(sd, dtype)
    return _finish(TR.swin_forward.__wrapped__(P, torch.as_tensor(x).to(dtype), **kw), labels, P)


def finite_difference(fn, sd, x, labels, name, index, h=1e-3):
    """Central difference of the loss w.r.t. element `index` (flat) of parameter `name` in fp64-ish steps: checks the autograd
    oracle itself (tests/test_oracle.py)."""
    out = []
    for sgn in (+1, -1):
        sd2 = dict(sd)
        a = np.array(sd[name], np.float64, copy=True)
        a.reshape(-1)[index] += sgn * h
        sd2[name] = a
        out.append(fn(sd2, x, labels)[0])
    return (out[0] - out[1]) / (2 * h)
