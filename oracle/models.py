"""CPU ORACLE (test infrastructure): numpy restatement of the four hot-path models of
paganpasta/eqxvision, single-sample like the reference, keyed by torch-style
`state_dict` names (see oracle/state.py).  PARITY UNPINNED -- see oracle/np_ops.py.

Each function cites the reference lines it follows.  `bf16=True` emulates the
product's storage precision (weights and inter-layer activations rounded to
bf16, fp32 accumulation, fp32 epilogues) so that kernel bugs are separated from
expected bf16 rounding.
"""
from __future__ import annotations

import numpy as np

from . import np_ops as O

F32 = np.float32


class _Q:
    """Optional bf16 storage emulation."""

    def __init__(self, bf16):
        self.on = bf16

    def __call__(self, x):
        return O.bf16_round(x) if self.on else np.asarray(x, F32)


def _bn_fold(sd, name, eps=1e-5):
    inv = 1.0 / np.sqrt(sd[name + ".running_var"].astype(F32) + F32(eps))
    scale = sd[name + ".weight"].astype(F32) * inv
    shift = sd[name + ".bias"].astype(F32) - sd[name + ".running_mean"].astype(F32) * scale
    return scale.astype(F32), shift.astype(F32)


def _conv_bn(sd, q, x, conv, bn, stride=1, padding=0, dilation=1, groups=1, relu=False, residual=None, eps=1e-5, act=None):
    """conv -> BN(inference) [-> + residual] [-> relu / `act`]; rounding only at the layer output."""
    y = O.conv2d(x, q(sd[conv + ".weight"]), None, stride, padding, dilation, groups)
    if bn is not None:
        if q.on:       # product folds BN into an fp32 scale/shift epilogue
            sc, sh = _bn_fold(sd, bn, eps)
            y = y * sc[:, None, None] + sh[:, None, None]
        else:          # literal reference order: (x-mean)/sqrt(var+eps)*w+b
            y = O.batchnorm_inference(y, sd[bn + ".weight"], sd[bn + ".bias"],
                                      sd[bn + ".running_mean"], sd[bn + ".running_var"], eps)
    if residual is not None:
        y = y + residual
    if relu:
        y = O.relu(y)
    if act is not None:
        y = act(y)
    return q(y)


# ---------------------------------------------------------------- alexnet.py:41-85
def alexnet_features(sd, x, bf16=False):
    q = _Q(bf16)
    x = q(x)

    def cr(x, name, stride, pad):
        y = O.conv2d(x, q(sd[name + ".weight"]), sd[name + ".bias"], stride, pad)
        return q(O.relu(y))

    x = cr(x, "features.0", 4, 2)
    x = O.maxpool2d(x, 3, 2)
    x = cr(x, "features.3", 1, 2)
    x = O.maxpool2d(x, 3, 2)
    x = cr(x, "features.6", 1, 1)
    x = cr(x, "features.8", 1, 1)
    x = cr(x, "features.10", 1, 1)
    x = O.maxpool2d(x, 3, 2)
    return x


def alexnet_forward(sd, x, bf16=False, key=None, dropout=0.5):
    """`key` given: TRAINING mode -- the classifier's two Dropout(p) layers (alexnet.py:63-68) draw from the keys
    nn.Sequential hands them: split(split(key, 2)[1], 7)[0] and [3] (alexnet.py:80-84)."""
    q = _Q(bf16)
    x = alexnet_features(sd, x, bf16)
    x = q(O.adaptive_avgpool2d(x, (6, 6)))                     # alexnet.py:82
    x = np.ravel(x)                                            # alexnet.py:83
    ks = None if key is None else O.jax_split(O.jax_split(key, 2)[1], 7)
    if ks is not None:
        x = q(O.dropout(x, dropout, ks[0]))
    x = q(O.relu(O.linear(x, q(sd["classifier.1.weight"]), sd["classifier.1.bias"])))
    if ks is not None:
        x = q(O.dropout(x, dropout, ks[3]))
    x = q(O.relu(O.linear(x, q(sd["classifier.4.weight"]), sd["classifier.4.bias"])))
    return O.linear(x, q(sd["classifier.6.weight"]), sd["classifier.6.bias"])


# ---------------------------------------------------------------- vgg.py:96-148
def vgg_forward(sd, x, plan, batch_norm=False, bf16=False):
    """features = [conv3x3(pad 1) (+BN) + relu | maxpool 2/2]*, AdaptiveAvgPool2d((7,7)), ravel, then the reference's
    classifier: Linear, Dropout(id), Linear, relu, Dropout(id), Linear -- ONE relu (vgg.py:96-105)."""
    q = _Q(bf16)
    x = q(x)
    i = 0
    for v in plan:
        if v == "M":
            x = O.maxpool2d(x, 2, 2)
            i += 1
            continue
        conv = f"features.{i}"
        i += 1
        if batch_norm:
            y = O.conv2d(x, q(sd[conv + ".weight"]), sd[conv + ".bias"], 1, 1)
            bn = f"features.{i}"
            i += 1
            y = O.batchnorm_inference(y, sd[bn + ".weight"], sd[bn + ".bias"], sd[bn + ".running_mean"], sd[bn + ".running_var"])
        else:
            y = O.conv2d(x, q(sd[conv + ".weight"]), sd[conv + ".bias"], 1, 1)
        x = q(O.relu(y))
        i += 1
    x = q(O.adaptive_avgpool2d(x, (7, 7)))
    x = np.ravel(x)
    x = q(O.linear(x, q(sd["classifier.0.weight"]), sd["classifier.0.bias"]))
    x = q(O.relu(O.linear(x, q(sd["classifier.3.weight"]), sd["classifier.3.bias"])))
    return O.linear(x, q(sd["classifier.6.weight"]), sd["classifier.6.bias"])


# ---------------------------------------------------------------- resnet.py:144-162, 335-358
def resnet_stages(sd, x, block="bottleneck", layers=(3, 4, 6, 3), bf16=False, groups=1, dilate=(False, False, False),
                  prefix=""):
    """Stem + the four stages; returns [layer1, layer2, layer3, layer4] outputs.  `dilate` = replace_stride_with_dilation
    (resnet.py:286-333: a dilated stage keeps stride 1, its first block uses the previous dilation, the others the new one)."""
    q = _Q(bf16)
    x = q(x)
    x = _conv_bn(sd, q, x, prefix + "conv1", prefix + "bn1", stride=2, padding=3, relu=True)   # resnet.py:344-346
    x = O.maxpool2d(x, 3, 2, 1)                                                      # resnet.py:347
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
            if (p + ".downsample.0.weight") in sd:                                   # resnet.py:295-303
                identity = _conv_bn(sd, q, x, p + ".downsample.0", p + ".downsample.1", stride=s)
            else:
                identity = x
            if block == "bottleneck":                                                # resnet.py:144-162
                out = _conv_bn(sd, q, x, p + ".conv1", p + ".bn1", relu=True)
                out = _conv_bn(sd, q, out, p + ".conv2", p + ".bn2", stride=s, padding=d, dilation=d, groups=groups, relu=True)
                x = _conv_bn(sd, q, out, p + ".conv3", p + ".bn3", relu=True, residual=identity)
            else:                                                                    # resnet.py:80-92
                out = _conv_bn(sd, q, x, p + ".conv1", p + ".bn1", stride=s, padding=1, relu=True)
                x = _conv_bn(sd, q, out, p + ".conv2", p + ".bn2", padding=1, relu=True, residual=identity)
        outs.append(x)
    return outs


def resnet_forward(sd, x, block="bottleneck", layers=(3, 4, 6, 3), bf16=False, groups=1):
    q = _Q(bf16)
    x = resnet_stages(sd, x, block, layers, bf16, groups)[-1]
    x = q(O.adaptive_avgpool2d(x, (1, 1)))                                           # resnet.py:354
    x = np.ravel(x)
    return O.linear(x, q(sd["fc.weight"]), sd["fc.bias"])                            # resnet.py:356


def resnet_forward_train(sd, xs, running, block="bottleneck", layers=(3, 4, 6, 3), bf16=False):
    """resnet_forward with every BatchNorm in TRAINING mode (np_ops.batchnorm_train), on a whole batch xs [B,3,H,W]: batch
    statistics couple the samples, so the layers run batch-wide.  `running`: dict bn-name -> (mean, var) or missing (first call);
    updated in place.  Returns logits [B, classes].  Data-parallel runs: pass the GLOBAL batch and slice the rank's rows."""
    q = _Q(bf16)

    def cbn(xs, conv, bn, stride=1, padding=0, relu=False, residual=None):
        ys = np.stack([O.conv2d(x, q(sd[conv + ".weight"]), None, stride, padding) for x in xs])
        ys = q(ys)                                              # the product materialises the convolution output
        ys, running[bn] = O.batchnorm_train(ys, sd[bn + ".weight"], sd[bn + ".bias"], running.get(bn), bn not in running)
        if residual is not None:
            ys = q(ys) + residual
        if relu:
            ys = O.relu(ys)
        return q(ys)

    xs = q(np.asarray(xs, np.float32))
    xs = cbn(xs, "conv1", "bn1", stride=2, padding=3, relu=True)
    xs = np.stack([O.maxpool2d(x, 3, 2, 1) for x in xs])
    for li, nblk in enumerate(layers):
        for bi in range(nblk):
            p = f"layer{li + 1}.{bi}"
            s = (1 if li == 0 else 2) if bi == 0 else 1
            identity = cbn(xs, p + ".downsample.0", p + ".downsample.1", stride=s) if (p + ".downsample.0.weight") in sd else xs
            if block == "bottleneck":
                out = cbn(xs, p + ".conv1", p + ".bn1", relu=True)
                out = cbn(out, p + ".conv2", p + ".bn2", stride=s, padding=1, relu=True)
                xs = cbn(out, p + ".conv3", p + ".bn3", relu=True, residual=identity)
            else:
                out = cbn(xs, p + ".conv1", p + ".bn1", stride=s, padding=1, relu=True)
                xs = cbn(out, p + ".conv2", p + ".bn2", padding=1, relu=True, residual=identity)
    feats = q(xs.mean(axis=(2, 3)))
    return np.stack([O.linear(f, q(sd["fc.weight"]), sd["fc.bias"]) for f in feats])


# ---------------------------------------------------------------- mobilenetv2.py:16-229
def mobilenet_v2_forward(sd, x, setting, bf16=False):
    """stem conv3x3/2 + BN + relu; inverted residuals ([1x1 expand + BN + relu,] 3x3 depthwise + BN + relu, 1x1 project + BN,
    + input when stride 1 and equal widths); 1x1 conv + BN + relu; global mean; Linear.  relu, not relu6 (mobilenetv2.py:54,66)."""
    q = _Q(bf16)
    x = _conv_bn(sd, q, q(x), "features.0.0", "features.0.1", stride=2, padding=1, relu=True)
    cin, i = x.shape[0], 1
    for t, c, n, s in setting:
        for r in range(n):
            stride = s if r == 0 else 1
            p, j = f"features.{i}.conv", 0
            h = x
            if t != 1:
                h = _conv_bn(sd, q, h, f"{p}.0.0", f"{p}.0.1", relu=True)
                j = 1
            hidden = h.shape[0]
            h = _conv_bn(sd, q, h, f"{p}.{j}.0", f"{p}.{j}.1", stride=stride, padding=1, groups=hidden, relu=True)
            res = x if (stride == 1 and cin == c) else None
            x = _conv_bn(sd, q, h, f"{p}.{j + 1}", f"{p}.{j + 2}", residual=res)
            cin = c
            i += 1
    x = _conv_bn(sd, q, x, f"features.{i}.0", f"features.{i}.1", relu=True)
    x = np.ravel(O.adaptive_avgpool2d(x, (1, 1)))
    return O.linear(x, q(sd["classifier.1.weight"]), sd["classifier.1.bias"])


# ---------------------------------------------------------------- mobilenetv3.py:46-245, layers/squeeze.py:11-61
def _se(sd, q, x, p):
    s = O.adaptive_avgpool2d(x, (1, 1))
    s = q(O.relu(O.conv2d(q(s), q(sd[p + ".fc1.weight"]), sd[p + ".fc1.bias"])))
    s = q(O.hard_sigmoid(O.conv2d(s, q(sd[p + ".fc2.weight"]), sd[p + ".fc2.bias"])))
    return q(x * s)


def mobilenet_v3_features(sd, x, conf, bf16=False, taps=(), prefix="features."):
    """The `features` stack; `conf` rows = (in, kernel, expanded, out, use_se, "RE"|"HS", stride, dilation) after channel
    adjustment; BatchNorm eps 1e-3 (mobilenetv3.py:188).  Returns (output, [outputs of the layers listed in `taps`])."""
    q = _Q(bf16)
    outs = {}
    x = _conv_bn(sd, q, q(x), prefix + "0.0", prefix + "0.1", stride=2, padding=1, eps=1e-3, act=O.hard_swish)
    if 0 in taps:
        outs[0] = x
    for i, (cin, k, cexp, cout, use_se, a, stride, dil) in enumerate(conf, start=1):
        act = O.hard_swish if a == "HS" else O.relu
        p, j, h = f"{prefix}{i}.block", 0, x
        if cexp != cin:
            h = _conv_bn(sd, q, h, f"{p}.0.0", f"{p}.0.1", eps=1e-3, act=act)
            j = 1
        h = _conv_bn(sd, q, h, f"{p}.{j}.0", f"{p}.{j}.1", stride=1 if dil > 1 else stride, padding=(k - 1) // 2 * dil, dilation=dil,
                     groups=cexp, eps=1e-3, act=act)
        j += 1
        if use_se:
            h = _se(sd, q, h, f"{p}.{j}")
            j += 1
        x = _conv_bn(sd, q, h, f"{p}.{j}.0", f"{p}.{j}.1", eps=1e-3, residual=x if (stride == 1 and cin == cout) else None)
        if i in taps:
            outs[i] = x
    i = len(conf) + 1
    x = _conv_bn(sd, q, x, f"{prefix}{i}.0", f"{prefix}{i}.1", eps=1e-3, act=O.hard_swish)
    if i in taps:
        outs[i] = x
    return x, [outs[t] for t in taps]


def mobilenet_v3_forward(sd, x, conf, bf16=False):
    q = _Q(bf16)
    x, _ = mobilenet_v3_features(sd, x, conf, bf16)
    x = q(np.ravel(O.adaptive_avgpool2d(x, (1, 1))))
    x = q(O.hard_swish(O.linear(x, q(sd["classifier.0.weight"]), sd["classifier.0.bias"])))
    return O.linear(x, q(sd["classifier.3.weight"]), sd["classifier.3.bias"])


# ---------------------------------------------------------------- regnet.py:15-400
def regnet_forward(sd, x, widths, depths, group_widths, se_ratio=0.25, bf16=False):
    """stem 3x3/2 + BN + relu; per block relu(proj(x) + c(se(b(a(x))))) with a = 1x1, b = 3x3 grouped (stride 2 in a stage's first
    block), se = squeeze-excitation (relu inside, sigmoid gate) for the Y variants, c = 1x1; global mean; fc."""
    q = _Q(bf16)
    x = _conv_bn(sd, q, q(x), "stem.0", "stem.1", stride=2, padding=1, relu=True)
    for si, (w, d, gw) in enumerate(zip(widths, depths, group_widths)):
        for b in range(d):
            p = f"trunk_output.{si}.{b}"
            stride = 2 if b == 0 else 1
            sc = _conv_bn(sd, q, x, p + ".proj.0", p + ".proj.1", stride=stride) if (p + ".proj.0.weight") in sd else x
            h = _conv_bn(sd, q, x, p + ".f.0.0", p + ".f.0.1", relu=True)
            h = _conv_bn(sd, q, h, p + ".f.1.0", p + ".f.1.1", stride=stride, padding=1, groups=w // gw, relu=True)
            j = 2
            if se_ratio:
                g = q(O.adaptive_avgpool2d(h, (1, 1)))
                g = q(O.relu(O.conv2d(g, q(sd[p + ".f.2.fc1.weight"]), sd[p + ".f.2.fc1.bias"])))
                g = q(O.sigmoid(O.conv2d(g, q(sd[p + ".f.2.fc2.weight"]), sd[p + ".f.2.fc2.bias"])))
                h = q(h * g)
                j = 3
            x = _conv_bn(sd, q, h, f"{p}.f.{j}.0", f"{p}.f.{j}.1", relu=True, residual=sc)
    x = np.ravel(O.adaptive_avgpool2d(x, (1, 1)))
    return O.linear(x, q(sd["fc.weight"]), sd["fc.bias"])


# ---------------------------------------------------------------- efficientnet.py:95-400
def efficientnet_forward(sd, x, stages, eps=1e-5, bf16=False):
    """stem 3x3/2 + BN + SiLU; per stage row (fused, expand, kernel, stride, in, out, layers): MBConv = [1x1 expand,] k x k depthwise,
    SE (SiLU inside, sigmoid gate, squeeze width max(1, in // 4)), 1x1 project; FusedMBConv = k x k conv [+ 1x1 project]; residual when
    stride 1 and in == out (stochastic depth = identity in inference); 1x1 conv + BN + SiLU; global mean; Linear."""
    q = _Q(bf16)
    md = lambda v: (lambda n: n + 8 if n < 0.9 * v else n)(max(8, int(v + 4) // 8 * 8))
    x = _conv_bn(sd, q, q(x), "features.0.0", "features.0.1", stride=2, padding=1, eps=eps, act=O.silu)
    for si, (fused, e, k, s, cin, cout, n) in enumerate(stages, start=1):
        for b in range(n):
            ci, stride = (cin, s) if b == 0 else (cout, 1)
            p = f"features.{si}.{b}.block"
            cexp = md(ci * e)
            res = x if (stride == 1 and ci == cout) else None
            if fused:
                if cexp != ci:
                    h = _conv_bn(sd, q, x, p + ".0.0", p + ".0.1", stride=stride, padding=(k - 1) // 2, eps=eps, act=O.silu)
                    x = _conv_bn(sd, q, h, p + ".1.0", p + ".1.1", eps=eps, residual=res)
                else:
                    h = _conv_bn(sd, q, x, p + ".0.0", p + ".0.1", stride=stride, padding=(k - 1) // 2, eps=eps, act=O.silu)
                    x = q(h + res) if res is not None else h
                continue
            j, h = 0, x
            if cexp != ci:
                h = _conv_bn(sd, q, h, f"{p}.0.0", f"{p}.0.1", eps=eps, act=O.silu)
                j = 1
            h = _conv_bn(sd, q, h, f"{p}.{j}.0", f"{p}.{j}.1", stride=stride, padding=(k - 1) // 2, groups=cexp, eps=eps, act=O.silu)
            se = f"{p}.{j + 1}"
            g = q(O.adaptive_avgpool2d(h, (1, 1)))
            g = q(O.silu(O.conv2d(g, q(sd[se + ".fc1.weight"]), sd[se + ".fc1.bias"])))
            g = q(O.sigmoid(O.conv2d(g, q(sd[se + ".fc2.weight"]), sd[se + ".fc2.bias"])))
            h = q(h * g)
            x = _conv_bn(sd, q, h, f"{p}.{j + 2}.0", f"{p}.{j + 2}.1", eps=eps, residual=res)
    i = len(stages) + 1
    x = _conv_bn(sd, q, x, f"features.{i}.0", f"features.{i}.1", eps=eps, act=O.silu)
    x = np.ravel(O.adaptive_avgpool2d(x, (1, 1)))
    return O.linear(x, q(sd["classifier.1.weight"]), sd["classifier.1.bias"])


# ---------------------------------------------------------------- lraspp.py:13-118
def lraspp_forward(sd, x, conf, taps=(4, 16), bf16=False):
    q = _Q(bf16)
    size = x.shape[-2:]
    _, (low, high) = mobilenet_v3_features(sd, x, conf, bf16, taps, prefix="backbone.")
    y = _conv_bn(sd, q, high, "classifier.cbr.0", "classifier.cbr.1", relu=True)
    s = q(O.sigmoid(O.conv2d(q(O.adaptive_avgpool2d(high, (1, 1))), q(sd["classifier.scale.1.weight"]))))
    y = q(O.resize_bilinear(q(y * s), low.shape[-2:]))
    out = q(O.conv2d(low, q(sd["classifier.low_classifier.weight"]), sd["classifier.low_classifier.bias"])) + \
        q(O.conv2d(y, q(sd["classifier.high_classifier.weight"]), sd["classifier.high_classifier.bias"]))
    return O.resize_bilinear(q(out), size)


# ---------------------------------------------------------------- segmentation/_utils.py:36-60, fcn.py:19-35, deeplabv3.py:24-136
def _fcn_head(sd, q, x, p):
    y = _conv_bn(sd, q, x, p + ".0", p + ".1", padding=1, relu=True)                 # conv3x3 + BN + relu (+ Dropout = id)
    return q(O.conv2d(y, q(sd[p + ".4.weight"]), sd[p + ".4.bias"]))


def _deeplab_head(sd, q, x, p):
    a = p + ".0"
    branches = [_conv_bn(sd, q, x, a + ".convs.0.0", a + ".convs.0.1", relu=True)]
    for i, rate in enumerate((12, 24, 36)):
        branches.append(_conv_bn(sd, q, x, f"{a}.convs.{i + 1}.0", f"{a}.convs.{i + 1}.1", padding=rate, dilation=rate, relu=True))
    pooled = q(O.adaptive_avgpool2d(x, (1, 1)))
    pooled = _conv_bn(sd, q, pooled, a + ".convs.4.1", a + ".convs.4.2", relu=True)
    branches.append(q(O.resize_bilinear(pooled, x.shape[-2:])))
    y = np.concatenate(branches, axis=0)
    y = _conv_bn(sd, q, y, a + ".project.0", a + ".project.1", relu=True)
    y = _conv_bn(sd, q, y, p + ".1", p + ".2", padding=1, relu=True)
    return q(O.conv2d(y, q(sd[p + ".4.weight"]), sd[p + ".4.bias"]))


def segmentation_forward(sd, x, kind="fcn", layers=(3, 4, 6, 3), aux=True, bf16=False):
    """(aux, out) of fcn / deeplabv3 on the dilated ResNet backbone ([False, True, True]); both at the input resolution."""
    q = _Q(bf16)
    size = x.shape[-2:]
    feats = resnet_stages(sd, x, "bottleneck", layers, bf16, dilate=(False, True, True), prefix="backbone.")
    head = _fcn_head if kind == "fcn" else _deeplab_head
    out = O.resize_bilinear(head(sd, q, feats[3], "classifier"), size)
    a = O.resize_bilinear(_fcn_head(sd, q, feats[2], "aux_classifier"), size) if aux else None
    return a, out


# ---------------------------------------------------------------- vit.py:139-157, 261-273
def vit_block(sd, q, x, p, num_heads, return_attention=False, drop_path=0.0, key=None, drop=0.0, attn_drop=0.0):
    """`key` given (training mode): keys = split(key, 4) (vit.py:148): [0] -> the attention, which splits it again (vit.py:63) for
    attn_drop on the (1, heads, N, N) probabilities (vit.py:71) and proj_drop = `drop` on its output (vit.py:75); [1] / [3] ->
    x + DropPath(y) (vit.py:153, 156; mode "global": one draw for the whole sample); [2] -> split(., N): one key per TOKEN of the
    vmapped MLP (vit.py:155), each split in two for its Dropouts after the activation and after fc2 (mlps.py:60-65)."""
    y = q(O.layernorm_rows(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]))       # vit.py:149
    N, C = y.shape
    dh = C // num_heads
    ks = None if key is None else O.jax_split(key, 4)
    aks = None if ks is None else O.jax_split(ks[0], 2)
    qkv = y @ q(sd[p + ".attn.qkv.weight"]).T
    if (p + ".attn.qkv.bias") in sd:
        qkv = qkv + sd[p + ".attn.qkv.bias"]
    qkv = q(qkv).reshape(N, 3, num_heads, dh).transpose(1, 2, 0, 3)                   # vit.py:65-66
    qq, kk, vv = qkv[0], qkv[1], qkv[2]
    attn = (qq @ np.transpose(kk, (0, 2, 1))) * F32(dh ** -0.5)                      # vit.py:69
    attn = O.softmax(attn, -1)                                                       # vit.py:70
    if aks is not None and attn_drop > 0.0:
        attn = O.dropout(attn[None], attn_drop, aks[0])[0]                           # vit.py:71
    if return_attention:
        return attn[None]                                                            # (1, heads, N, N)
    a = q(np.transpose(q(attn) @ vv if q.on else attn @ vv, (1, 0, 2)).reshape(N, C))  # vit.py:73
    y = a @ q(sd[p + ".attn.proj.weight"]).T + sd[p + ".attn.proj.bias"]             # vit.py:74
    if aks is not None and drop > 0.0:
        y = O.dropout(q(y), drop, aks[1])                                            # vit.py:75
    if ks is not None and drop_path > 0.0:
        y = O.drop_path(q(y), drop_path, "global", ks[1])
    x = q(x + y)                                                                     # vit.py:153
    y = q(O.layernorm_rows(x, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]))       # vit.py:154
    h = q(O.gelu_tanh(y @ q(sd[p + ".mlp.fc1.weight"]).T + sd[p + ".mlp.fc1.bias"]))  # mlps.py:61-62
    mks = None
    if ks is not None and drop > 0.0:
        mks = [O.jax_split(tk, 2) for tk in O.jax_split(ks[2], N)]                   # vit.py:155, mlps.py:60
        h = q(np.stack([O.dropout(h[t], drop, mks[t][0]) for t in range(N)]))        # mlps.py:63
    y = h @ q(sd[p + ".mlp.fc2.weight"]).T + sd[p + ".mlp.fc2.bias"]                 # mlps.py:64
    if mks is not None:
        y = np.stack([O.dropout(q(y[t]), drop, mks[t][1]) for t in range(N)])        # mlps.py:65
    if ks is not None and drop_path > 0.0:
        y = O.drop_path(q(y), drop_path, "global", ks[3])
    return q(x + y)                                                                  # vit.py:156


def vit_tokens(sd, q, x, patch):
    x = q(x)
    t = O.patch_embed(x, q(sd["patch_embed.proj.weight"]), sd["patch_embed.proj.bias"], patch)  # vit.py:268
    D = t.shape[1]
    cls = sd["cls_token"].reshape(1, D)
    pos = sd["pos_embed"].reshape(-1, D)
    return q(np.concatenate([cls, t], 0) + pos)                                      # vit.py:269


def vit_forward(sd, x, patch=16, num_heads=12, depth=12, bf16=False, key=None, drop_path_rate=0.0, drop_rate=0.0,
                attn_drop_rate=0.0):
    """`key` given: TRAINING mode: block i drops its paths with linspace(0, drop_path_rate, depth)[i] (vit.py:236-246) and runs
    its Dropouts (drop_rate: projection + MLP, attn_drop_rate: attention probabilities) from split(key, depth)[i] (vit.py:267-271).
    (`pos_drop` is constructed but never applied by the reference's __call__.)"""
    q = _Q(bf16)
    x = vit_tokens(sd, q, x, patch)
    bkeys = None if key is None else O.jax_split(key, depth)
    dpr = np.linspace(0, drop_path_rate, depth)
    for i in range(depth):
        x = vit_block(sd, q, x, f"blocks.{i}", num_heads, drop_path=float(dpr[i]), key=None if bkeys is None else bkeys[i],
                      drop=drop_rate, attn_drop=attn_drop_rate)
    x = q(O.layernorm_rows(x, sd["norm.weight"], sd["norm.bias"]))                   # vit.py:272
    if "fc.weight" in sd:
        return O.linear(x[0], q(sd["fc.weight"]), sd["fc.bias"])                     # vit.py:273
    return x[0]


def vit_last_self_attention(sd, x, patch=16, num_heads=12, depth=12, bf16=False):
    """vit.py:275-292"""
    q = _Q(bf16)
    x = vit_tokens(sd, q, x, patch)
    for i in range(depth - 1):
        x = vit_block(sd, q, x, f"blocks.{i}", num_heads)
    return vit_block(sd, q, x, f"blocks.{depth - 1}", num_heads, return_attention=True)


# ---------------------------------------------------------------- swin.py:572-578, 760-772
def swin_block(sd, q, x, p, num_heads, window, shift, sd_prob=0.0, key=None, dropout=0.0, attention_dropout=0.0, training=True):
    """`key` given: keys = split(key, 4) (swin.py:573): [0] -> the attention's two `_func_dropout` draws (every mode), [2] -> the
    MLP (split in two for its Dropouts on the (H, W, C) arrays, mlps.py:60-65; training mode only), [1] / [3] ->
    x + stochastic_depth(., key) with DropPath(sd_prob, mode="local"): one draw per channel (swin.py:545, 572-578; training mode)."""
    ks = None if key is None else O.jax_split(key, 4)
    y = q(O.layernorm2d(x, sd[p + ".norm1.weight"], sd[p + ".norm1.bias"]))
    bias = O.relative_position_bias(sd[p + ".attn.relative_position_bias_table"],
                                    sd[p + ".attn.relative_position_index"], window)
    y = O.shifted_window_attention(y, q(sd[p + ".attn.qkv.weight"]), q(sd[p + ".attn.proj.weight"]), bias,
                                   window, num_heads, shift, sd[p + ".attn.qkv.bias"], sd[p + ".attn.proj.bias"],
                                   attention_dropout=attention_dropout, dropout=dropout, key=None if ks is None else ks[0])
    if ks is not None and training and sd_prob > 0.0:
        y = O.drop_path(q(y), sd_prob, "local", ks[1])
    x = q(x + y)
    y = q(O.layernorm2d(x, sd[p + ".norm2.weight"], sd[p + ".norm2.bias"]))
    h = q(O.gelu_tanh(O.linear2d(y, q(sd[p + ".mlp.0.weight"]), sd[p + ".mlp.0.bias"])))
    mk = O.jax_split(ks[2], 2) if ks is not None and training and dropout > 0.0 else None
    # Linear2d returns (out_features, h, w) (extensions_2d.py:46-50): eqx.nn.Dropout sees (C, H, W) arrays here
    if mk is not None:
        h = q(O.dropout(h, dropout, mk[0]))
    y = O.linear2d(h, q(sd[p + ".mlp.3.weight"]), sd[p + ".mlp.3.bias"])
    if mk is not None:
        y = O.dropout(q(y), dropout, mk[1])
    if ks is not None and training and sd_prob > 0.0:
        y = O.drop_path(q(y), sd_prob, "local", ks[3])
    return q(x + y)


def swin_forward(sd, x, patch=(4, 4), depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24), window=(7, 7), bf16=False, key=None,
                 stochastic_depth_prob=0.0, dropout=0.0, attention_dropout=0.0, training=True):
    """`key` given: TRAINING mode.  Keys as the reference derives them: split(key, 2)[0] for `features` (swin.py:766-767),
    nn.Sequential splits it per layer, each stage Sequential per block; block id / (total - 1) scales the drop rate (:730-733)."""
    q = _Q(bf16)
    x = q(x)
    nfeat = 2 * len(depths)                                   # patch embedding, then stage / merge alternating
    fkeys = None if key is None else O.jax_split(O.jax_split(key, 2)[0], nfeat)
    total, blk_id = sum(depths), 0
    x = O.conv2d(x, q(sd["features.0.0.weight"]), sd["features.0.0.bias"], stride=tuple(patch))   # swin.py:705-711
    x = q(O.layernorm2d(q(x), sd["features.0.2.weight"], sd["features.0.2.bias"]))
    fi = 1
    for si, depth in enumerate(depths):
        bkeys = None if fkeys is None else O.jax_split(fkeys[fi], depth)
        for bi in range(depth):
            shift = [0 if bi % 2 == 0 else w // 2 for w in window]                    # swin.py:736-738
            sdp = stochastic_depth_prob * float(blk_id) / (total - 1) if total > 1 else 0.0
            x = swin_block(sd, q, x, f"features.{fi}.{bi}", num_heads[si], list(window), shift, sdp,
                           None if bkeys is None else bkeys[bi], dropout, attention_dropout, training)
            blk_id += 1
        fi += 1
        if si < len(depths) - 1:
            p = f"features.{fi}"
            x = q(O.patch_merging(x, sd[p + ".norm.weight"], sd[p + ".norm.bias"], q(sd[p + ".reduction.weight"])))
            fi += 1
    x = q(O.layernorm2d(x, sd["norm.weight"], sd["norm.bias"]))                      # swin.py:768
    x = q(O.adaptive_avgpool2d(x, (1, 1)))                                           # swin.py:769
    return O.linear(np.ravel(x), q(sd["head.weight"]), sd["head.bias"])              # swin.py:771
