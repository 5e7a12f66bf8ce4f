"""CPU ORACLE (test infrastructure): an INDEPENDENT batched fp32 restatement of the same four
models on `torch.nn.functional` (CPU).  Two jobs:

  1. cross-check of `oracle/models.py` (numpy): the two must agree to <=1e-4 relative;
     torchvision semantics are what the reference's own tests use as ground truth
     (`/root/reference/tests/test_models/test_resnet.py:24`).
  2. the `cpu_baseline` leg of bench.py ("port": CPU restatement, not JAX -- jax/equinox are
     not installed on the box), timed on all host cores.

PARITY UNPINNED -- see oracle/np_ops.py.  Never imported by the product.
"""
from __future__ import annotations

import numpy as np
import torch
import torch.nn.functional as F


def _t(sd):
    out = {}
    for k, v in sd.items():
        t = torch.from_numpy(np.ascontiguousarray(v)) if isinstance(v, np.ndarray) else v
        if isinstance(t, torch.Tensor) and t.dim() == 3 and tuple(t.shape[1:]) == (1, 1):
            t = t.reshape(-1)            # equinox stores conv biases as (C,1,1); torch wants (C,)
        out[k] = t
    return out


def _bn(sd, x, name, eps=1e-5):
    return F.batch_norm(x, sd[name + ".running_mean"], sd[name + ".running_var"],
                        sd[name + ".weight"], sd[name + ".bias"], False, 0.0, eps)


@torch.no_grad()
def alexnet_features(sd, x):
    sd = _t(sd)
    x = torch.as_tensor(x)
    x = F.relu(F.conv2d(x, sd["features.0.weight"], sd["features.0.bias"], 4, 2))
    x = F.max_pool2d(x, 3, 2)
    x = F.relu(F.conv2d(x, sd["features.3.weight"], sd["features.3.bias"], 1, 2))
    x = F.max_pool2d(x, 3, 2)
    x = F.relu(F.conv2d(x, sd["features.6.weight"], sd["features.6.bias"], 1, 1))
    x = F.relu(F.conv2d(x, sd["features.8.weight"], sd["features.8.bias"], 1, 1))
    x = F.relu(F.conv2d(x, sd["features.10.weight"], sd["features.10.bias"], 1, 1))
    return F.max_pool2d(x, 3, 2)


@torch.no_grad()
def alexnet_forward(sd, x):
    t = _t(sd)
    x = alexnet_features(sd, x)
    x = F.adaptive_avg_pool2d(x, (6, 6)).flatten(1)
    x = F.relu(F.linear(x, t["classifier.1.weight"], t["classifier.1.bias"]))
    x = F.relu(F.linear(x, t["classifier.4.weight"], t["classifier.4.bias"]))
    return F.linear(x, t["classifier.6.weight"], t["classifier.6.bias"])


@torch.no_grad()
def vgg_forward(sd, x, plan, batch_norm=False):
    """torch.nn.functional restatement of the reference VGG (vgg.py:96-148), incl. its single-relu classifier.  Only valid
    where equinox's AdaptiveAvgPool2d agrees with torch's (input size divisible by 7 after the pools)."""
    t = _t(sd)
    x = torch.as_tensor(x)
    i = 0
    for v in plan:
        if v == "M":
            x = F.max_pool2d(x, 2, 2)
            i += 1
            continue
        x = F.conv2d(x, t[f"features.{i}.weight"], t[f"features.{i}.bias"], 1, 1)
        i += 1
        if batch_norm:
            x = _bn(t, x, f"features.{i}")
            i += 1
        x = F.relu(x)
        i += 1
    assert x.shape[-1] % 7 == 0 and x.shape[-2] % 7 == 0
    x = F.adaptive_avg_pool2d(x, (7, 7)).flatten(1)
    x = F.linear(x, t["classifier.0.weight"], t["classifier.0.bias"])
    x = F.relu(F.linear(x, t["classifier.3.weight"], t["classifier.3.bias"]))
    return F.linear(x, t["classifier.6.weight"], t["classifier.6.bias"])


def _resnet_stages(sd, x, block="bottleneck", layers=(3, 4, 6, 3), groups=1, dilate=(False, False, False), prefix=""):
    x = F.relu(_bn(sd, F.conv2d(x, sd[prefix + "conv1.weight"], None, 2, 3), prefix + "bn1"))
    x = F.max_pool2d(x, 3, 2, 1)
    outs, dilation = [], 1
    for li, nblk in enumerate(layers):
        stride = 1 if li == 0 else 2
        prev = dilation
        if li > 0 and dilate[li - 1]:
            dilation *= stride
            stride = 1
        for bi in range(nblk):
            p = f"{prefix}layer{li + 1}.{bi}"
            s = stride if bi == 0 else 1
            d = prev if bi == 0 else dilation
            idt = x
            if (p + ".downsample.0.weight") in sd:
                idt = _bn(sd, F.conv2d(x, sd[p + ".downsample.0.weight"], None, s), p + ".downsample.1")
            if block == "bottleneck":
                o = F.relu(_bn(sd, F.conv2d(x, sd[p + ".conv1.weight"]), p + ".bn1"))
                o = F.relu(_bn(sd, F.conv2d(o, sd[p + ".conv2.weight"], None, s, d, d, groups), p + ".bn2"))
                o = _bn(sd, F.conv2d(o, sd[p + ".conv3.weight"]), p + ".bn3")
            else:
                o = F.relu(_bn(sd, F.conv2d(x, sd[p + ".conv1.weight"], None, s, 1), p + ".bn1"))
                o = _bn(sd, F.conv2d(o, sd[p + ".conv2.weight"], None, 1, 1), p + ".bn2")
            x = F.relu(o + idt)
        outs.append(x)
    return outs


@torch.no_grad()
def resnet_forward(sd, x, block="bottleneck", layers=(3, 4, 6, 3), groups=1):
    sd = _t(sd)
    x = _resnet_stages(sd, torch.as_tensor(x), block, layers, groups)[-1]
    x = F.adaptive_avg_pool2d(x, 1).flatten(1)
    return F.linear(x, sd["fc.weight"], sd["fc.bias"])


@torch.no_grad()
def mobilenet_v2_forward(sd, x, setting):
    sd = _t(sd)
    x = torch.as_tensor(x)
    cbr = lambda x, c, b, stride=1, pad=0, groups=1, relu=True: (lambda y: F.relu(y) if relu else y)(
        _bn(sd, F.conv2d(x, sd[c + ".weight"], None, stride, pad, 1, groups), b))
    x = cbr(x, "features.0.0", "features.0.1", 2, 1)
    cin, i = x.shape[1], 1
    for t, c, n, s in setting:
        for r in range(n):
            stride = s if r == 0 else 1
            p, j = f"features.{i}.conv", 0
            h = x
            if t != 1:
                h = cbr(h, f"{p}.0.0", f"{p}.0.1")
                j = 1
            h = cbr(h, f"{p}.{j}.0", f"{p}.{j}.1", stride, 1, h.shape[1])
            y = cbr(h, f"{p}.{j + 1}", f"{p}.{j + 2}", relu=False)
            x = x + y if (stride == 1 and cin == c) else y
            cin = c
            i += 1
    x = cbr(x, f"features.{i}.0", f"features.{i}.1")
    x = x.mean((2, 3))
    return F.linear(x, sd["classifier.1.weight"], sd["classifier.1.bias"])


def _mbv3_features_t(sd, x, conf, taps=(), prefix="features."):
    outs = {}
    hs = F.hardswish
    cba = lambda x, c, b, act, stride=1, pad=0, dil=1, groups=1: (lambda y: y if act is None else act(y))(
        _bn(sd, F.conv2d(x, sd[c + ".weight"], None, stride, pad, dil, groups), b, 1e-3))
    x = cba(x, prefix + "0.0", prefix + "0.1", hs, 2, 1)
    for i, (cin, k, cexp, cout, use_se, a, stride, dil) in enumerate(conf, start=1):
        act = hs if a == "HS" else F.relu
        p, j, h = f"{prefix}{i}.block", 0, x
        if cexp != cin:
            h = cba(h, f"{p}.0.0", f"{p}.0.1", act)
            j = 1
        h = cba(h, f"{p}.{j}.0", f"{p}.{j}.1", act, 1 if dil > 1 else stride, (k - 1) // 2 * dil, dil, cexp)
        j += 1
        if use_se:
            s = F.adaptive_avg_pool2d(h, 1)
            s = F.relu(F.conv2d(s, sd[f"{p}.{j}.fc1.weight"], sd[f"{p}.{j}.fc1.bias"]))
            s = F.hardsigmoid(F.conv2d(s, sd[f"{p}.{j}.fc2.weight"], sd[f"{p}.{j}.fc2.bias"]))
            h = h * s
            j += 1
        y = cba(h, f"{p}.{j}.0", f"{p}.{j}.1", None)
        x = x + y if (stride == 1 and cin == cout) else y
        if i in taps:
            outs[i] = x
    i = len(conf) + 1
    x = cba(x, f"{prefix}{i}.0", f"{prefix}{i}.1", hs)
    if i in taps:
        outs[i] = x
    return x, [outs[t] for t in taps]


@torch.no_grad()
def mobilenet_v3_forward(sd, x, conf):
    sd = _t(sd)
    x, _ = _mbv3_features_t(sd, torch.as_tensor(x), conf)
    x = x.mean((2, 3))
    x = F.hardswish(F.linear(x, sd["classifier.0.weight"], sd["classifier.0.bias"]))
    return F.linear(x, sd["classifier.3.weight"], sd["classifier.3.bias"])


@torch.no_grad()
def lraspp_forward(sd, x, conf, taps=(4, 16)):
    sd = _t(sd)
    x = torch.as_tensor(x)
    size = x.shape[-2:]
    _, (low, high) = _mbv3_features_t(sd, x, conf, taps, prefix="backbone.")
    y = F.relu(_bn(sd, F.conv2d(high, sd["classifier.cbr.0.weight"]), "classifier.cbr.1"))
    s = torch.sigmoid(F.conv2d(F.adaptive_avg_pool2d(high, 1), sd["classifier.scale.1.weight"]))
    y = F.interpolate(y * s, size=low.shape[-2:], mode="bilinear", align_corners=False)
    out = F.conv2d(low, sd["classifier.low_classifier.weight"], sd["classifier.low_classifier.bias"]) + \
        F.conv2d(y, sd["classifier.high_classifier.weight"], sd["classifier.high_classifier.bias"])
    return F.interpolate(out, size=size, mode="bilinear", align_corners=False)


@torch.no_grad()
def efficientnet_forward(sd, x, stages, eps=1e-5):
    sd = _t(sd)
    x = torch.as_tensor(x)
    md = lambda v: (lambda n: n + 8 if n < 0.9 * v else n)(max(8, int(v + 4) // 8 * 8))
    cba = lambda x, c, b, act, stride=1, pad=0, groups=1: (lambda y: y if act is None else act(y))(
        _bn(sd, F.conv2d(x, sd[c + ".weight"], None, stride, pad, 1, groups), b, eps))
    x = cba(x, "features.0.0", "features.0.1", F.silu, 2, 1)
    for si, (fused, e, k, s, cin, cout, n) in enumerate(stages, start=1):
        for b in range(n):
            ci, stride = (cin, s) if b == 0 else (cout, 1)
            p = f"features.{si}.{b}.block"
            cexp = md(ci * e)
            use_res = stride == 1 and ci == cout
            if fused:
                h = cba(x, p + ".0.0", p + ".0.1", F.silu, stride, (k - 1) // 2)
                y = cba(h, p + ".1.0", p + ".1.1", None) if cexp != ci else h
            else:
                j, h = 0, x
                if cexp != ci:
                    h = cba(h, f"{p}.0.0", f"{p}.0.1", F.silu)
                    j = 1
                h = cba(h, f"{p}.{j}.0", f"{p}.{j}.1", F.silu, stride, (k - 1) // 2, cexp)
                se = f"{p}.{j + 1}"
                g = F.adaptive_avg_pool2d(h, 1)
                g = F.silu(F.conv2d(g, sd[se + ".fc1.weight"], sd[se + ".fc1.bias"]))
                g = torch.sigmoid(F.conv2d(g, sd[se + ".fc2.weight"], sd[se + ".fc2.bias"]))
                y = cba(h * g, f"{p}.{j + 2}.0", f"{p}.{j + 2}.1", None)
            x = x + y if use_res else y
    i = len(stages) + 1
    x = cba(x, f"features.{i}.0", f"features.{i}.1", F.silu)
    return F.linear(x.mean((2, 3)), sd["classifier.1.weight"], sd["classifier.1.bias"])


@torch.no_grad()
def regnet_forward(sd, x, widths, depths, group_widths, se_ratio=0.25):
    sd = _t(sd)
    x = torch.as_tensor(x)
    cb = lambda x, c, b, stride=1, pad=0, groups=1: _bn(sd, F.conv2d(x, sd[c + ".weight"], None, stride, pad, 1, groups), b)
    x = F.relu(cb(x, "stem.0", "stem.1", 2, 1))
    for si, (w, d, gw) in enumerate(zip(widths, depths, group_widths)):
        for b in range(d):
            p = f"trunk_output.{si}.{b}"
            stride = 2 if b == 0 else 1
            sc = cb(x, p + ".proj.0", p + ".proj.1", stride) if (p + ".proj.0.weight") in sd else x
            h = F.relu(cb(x, p + ".f.0.0", p + ".f.0.1"))
            h = F.relu(cb(h, p + ".f.1.0", p + ".f.1.1", stride, 1, w // gw))
            j = 2
            if se_ratio:
                g = F.adaptive_avg_pool2d(h, 1)
                g = F.relu(F.conv2d(g, sd[p + ".f.2.fc1.weight"], sd[p + ".f.2.fc1.bias"]))
                h = h * torch.sigmoid(F.conv2d(g, sd[p + ".f.2.fc2.weight"], sd[p + ".f.2.fc2.bias"]))
                j = 3
            x = F.relu(sc + cb(h, f"{p}.f.{j}.0", f"{p}.f.{j}.1"))
    return F.linear(x.mean((2, 3)), sd["fc.weight"], sd["fc.bias"])


def _fcn_head_t(sd, x, p):
    y = F.relu(_bn(sd, F.conv2d(x, sd[p + ".0.weight"], None, 1, 1), p + ".1"))
    return F.conv2d(y, sd[p + ".4.weight"], sd[p + ".4.bias"])


def _deeplab_head_t(sd, x, p):
    a = p + ".0"
    br = [F.relu(_bn(sd, F.conv2d(x, sd[a + ".convs.0.0.weight"]), a + ".convs.0.1"))]
    for i, r in enumerate((12, 24, 36)):
        br.append(F.relu(_bn(sd, F.conv2d(x, sd[f"{a}.convs.{i + 1}.0.weight"], None, 1, r, r), f"{a}.convs.{i + 1}.1")))
    g = F.adaptive_avg_pool2d(x, 1)
    g = F.relu(_bn(sd, F.conv2d(g, sd[a + ".convs.4.1.weight"]), a + ".convs.4.2"))
    br.append(F.interpolate(g, size=x.shape[-2:], mode="bilinear", align_corners=False))
    y = torch.cat(br, 1)
    y = F.relu(_bn(sd, F.conv2d(y, sd[a + ".project.0.weight"]), a + ".project.1"))
    y = F.relu(_bn(sd, F.conv2d(y, sd[p + ".1.weight"], None, 1, 1), p + ".2"))
    return F.conv2d(y, sd[p + ".4.weight"], sd[p + ".4.bias"])


@torch.no_grad()
def segmentation_forward(sd, x, kind="fcn", layers=(3, 4, 6, 3), aux=True):
    """torch.nn.functional restatement of fcn / deeplabv3 on the dilated ResNet ([False, True, True]); up-sampling with
    align_corners=False == jax.image.resize "bilinear" for out >= in (checked in tests/test_oracle.py)."""
    sd = _t(sd)
    x = torch.as_tensor(x)
    size = x.shape[-2:]
    feats = _resnet_stages(sd, x, "bottleneck", layers, dilate=(False, True, True), prefix="backbone.")
    head = _fcn_head_t if kind == "fcn" else _deeplab_head_t
    out = F.interpolate(head(sd, feats[3], "classifier"), size=size, mode="bilinear", align_corners=False)
    a = None
    if aux:
        a = F.interpolate(_fcn_head_t(sd, feats[2], "aux_classifier"), size=size, mode="bilinear", align_corners=False)
    return a, out


def _vit_tokens(sd, x, patch):
    x = F.conv2d(x, sd["patch_embed.proj.weight"], sd["patch_embed.proj.bias"], patch)
    x = x.flatten(2).transpose(1, 2)
    D = x.shape[-1]
    cls = sd["cls_token"].reshape(1, 1, D).expand(x.shape[0], -1, -1)
    return torch.cat([cls, x], 1) + sd["pos_embed"].reshape(1, -1, D)


def _vit_block(sd, x, p, H, return_attention=False):
    B, N, C = x.shape
    y = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
    qkv = F.linear(y, sd[p + ".attn.qkv.weight"], sd.get(p + ".attn.qkv.bias"))
    qkv = qkv.reshape(B, N, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0], qkv[1], qkv[2]
    attn = ((q @ k.transpose(-2, -1)) * (C // H) ** -0.5).softmax(-1)
    if return_attention:
        return attn[:, None]
    y = (attn @ v).transpose(1, 2).reshape(B, N, C)
    x = x + F.linear(y, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    y = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
    y = F.gelu(F.linear(y, sd[p + ".mlp.fc1.weight"], sd[p + ".mlp.fc1.bias"]), approximate="tanh")
    return x + F.linear(y, sd[p + ".mlp.fc2.weight"], sd[p + ".mlp.fc2.bias"])


@torch.no_grad()
def vit_forward(sd, x, patch=16, num_heads=12, depth=12):
    sd = _t(sd)
    x = _vit_tokens(sd, torch.as_tensor(x), patch)
    for i in range(depth):
        x = _vit_block(sd, x, f"blocks.{i}", num_heads)
    x = F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-5)[:, 0]
    if "fc.weight" in sd:
        x = F.linear(x, sd["fc.weight"], sd["fc.bias"])
    return x


@torch.no_grad()
def vit_last_self_attention(sd, x, patch=16, num_heads=12, depth=12):
    sd = _t(sd)
    x = _vit_tokens(sd, torch.as_tensor(x), patch)
    for i in range(depth - 1):
        x = _vit_block(sd, x, f"blocks.{i}", num_heads)
    return _vit_block(sd, x, f"blocks.{depth - 1}", num_heads, return_attention=True)


def _swin_attn(sd, x, p, H, ws, shift):
    """torchvision-style shifted_window_attention on NHWC (independent of the numpy oracle)."""
    B, Hh, Ww, C = x.shape
    shift = list(shift)
    if ws[0] >= Hh:
        shift[0] = 0
    if ws[1] >= Ww:
        shift[1] = 0
    if sum(shift) > 0:
        x = torch.roll(x, (-shift[0], -shift[1]), (1, 2))
    nW = (Hh // ws[0]) * (Ww // ws[1])
    n = ws[0] * ws[1]
    x = x.view(B, Hh // ws[0], ws[0], Ww // ws[1], ws[1], C).permute(0, 1, 3, 2, 4, 5).reshape(B * nW, n, C)
    qkv = F.linear(x, sd[p + ".attn.qkv.weight"], sd[p + ".attn.qkv.bias"])
    qkv = qkv.reshape(B * nW, n, 3, H, C // H).permute(2, 0, 3, 1, 4)
    q, k, v = qkv[0] * (C // H) ** -0.5, qkv[1], qkv[2]
    attn = q @ k.transpose(-2, -1)
    table = sd[p + ".attn.relative_position_bias_table"]
    idx = sd[p + ".attn.relative_position_index"].long().view(-1)
    bias = table[idx].view(n, n, -1).permute(2, 0, 1)
    attn = attn + bias[None]
    if sum(shift) > 0:
        m = x.new_zeros((Hh, Ww))
        hs = ((0, -ws[0]), (-ws[0], -shift[0]), (-shift[0], None))
        wsl = ((0, -ws[1]), (-ws[1], -shift[1]), (-shift[1], None))
        c = 0
        for h in hs:
            for w in wsl:
                m[h[0]:h[1], w[0]:w[1]] = c
                c += 1
        m = m.view(Hh // ws[0], ws[0], Ww // ws[1], ws[1]).permute(0, 2, 1, 3).reshape(nW, n)
        m = m.unsqueeze(1) - m.unsqueeze(2)
        m = m.masked_fill(m != 0, -100.0).masked_fill(m == 0, 0.0)
        attn = attn.view(B, nW, H, n, n) + m[None, :, None]
        attn = attn.view(-1, H, n, n)
    attn = attn.softmax(-1)
    y = (attn @ v).transpose(1, 2).reshape(B * nW, n, C)
    y = F.linear(y, sd[p + ".attn.proj.weight"], sd[p + ".attn.proj.bias"])
    y = y.view(B, Hh // ws[0], Ww // ws[1], ws[0], ws[1], C).permute(0, 1, 3, 2, 4, 5).reshape(B, Hh, Ww, C)
    if sum(shift) > 0:
        y = torch.roll(y, (shift[0], shift[1]), (1, 2))
    return y


@torch.no_grad()
def swin_forward(sd, x, patch=(4, 4), depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window=(7, 7)):
    sd = _t(sd)
    x = torch.as_tensor(x)
    x = F.conv2d(x, sd["features.0.0.weight"], sd["features.0.0.bias"], tuple(patch)).permute(0, 2, 3, 1)
    x = F.layer_norm(x, (x.shape[-1],), sd["features.0.2.weight"], sd["features.0.2.bias"], 1e-5)
    fi = 1
    for si, depth in enumerate(depths):
        for bi in range(depth):
            p = f"features.{fi}.{bi}"
            C = x.shape[-1]
            shift = [0 if bi % 2 == 0 else w // 2 for w in window]
            y = F.layer_norm(x, (C,), sd[p + ".norm1.weight"], sd[p + ".norm1.bias"], 1e-5)
            x = x + _swin_attn(sd, y, p, num_heads[si], list(window), shift)
            y = F.layer_norm(x, (C,), sd[p + ".norm2.weight"], sd[p + ".norm2.bias"], 1e-5)
            y = F.gelu(F.linear(y, sd[p + ".mlp.0.weight"], sd[p + ".mlp.0.bias"]), approximate="tanh")
            x = x + F.linear(y, sd[p + ".mlp.3.weight"], sd[p + ".mlp.3.bias"])
        fi += 1
        if si < len(depths) - 1:
            p = f"features.{fi}"
            x0, x1, x2, x3 = x[:, 0::2, 0::2], x[:, 1::2, 0::2], x[:, 0::2, 1::2], x[:, 1::2, 1::2]
            x = torch.cat([x0, x1, x2, x3], -1)
            x = F.layer_norm(x, (x.shape[-1],), sd[p + ".norm.weight"], sd[p + ".norm.bias"], 1e-5)
            x = F.linear(x, sd[p + ".reduction.weight"])
            fi += 1
    x = F.layer_norm(x, (x.shape[-1],), sd["norm.weight"], sd["norm.bias"], 1e-5)
    x = x.mean((1, 2))
    return F.linear(x, sd["head.weight"], sd["head.bias"])
