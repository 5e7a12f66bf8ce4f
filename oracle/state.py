"""CPU ORACLE support (test infrastructure): synthetic torch-style `state_dict`s.

The reference loads weights by ORDER from a torchvision / DINO `state_dict`
(`eqxvision/utils.py:172-199`), so the synthetic checkpoints below follow the
torchvision registration order and key names.  The same dict is fed (a) to the
product through `load_torch_weights` and (b) to the oracle models in
`oracle/models.py`, which is how both sides see identical weights.

Distributions follow SURVEY.md section 8(d): Conv/Linear U(+-1/sqrt(fan_in)) (equinox default
init), BN gamma~U[0.5,1.5], beta~N(0,0.1), mean~N(0,0.1), var~U[0.5,1.5].
"""
from __future__ import annotations

from collections import OrderedDict

import numpy as np

F32 = np.float32


def _u(rng, shape, fan_in):
    lim = 1.0 / np.sqrt(fan_in)
    return rng.uniform(-lim, lim, size=shape).astype(F32)


def _conv(sd, rng, name, cin, cout, k, bias, groups=1):
    kh, kw = (k, k) if isinstance(k, int) else k
    fan = cin // groups * kh * kw
    sd[name + ".weight"] = _u(rng, (cout, cin // groups, kh, kw), fan)
    if bias:
        sd[name + ".bias"] = _u(rng, (cout,), fan)


def _linear(sd, rng, name, fin, fout, bias=True):
    sd[name + ".weight"] = _u(rng, (fout, fin), fin)
    if bias:
        sd[name + ".bias"] = _u(rng, (fout,), fin)


def _bn(sd, rng, name, c):
    sd[name + ".weight"] = rng.uniform(0.5, 1.5, c).astype(F32)
    sd[name + ".bias"] = (0.1 * rng.standard_normal(c)).astype(F32)
    sd[name + ".running_mean"] = (0.1 * rng.standard_normal(c)).astype(F32)
    sd[name + ".running_var"] = rng.uniform(0.5, 1.5, c).astype(F32)
    sd[name + ".num_batches_tracked"] = np.asarray(1, np.int64)


def _ln(sd, rng, name, c, randomize=True):
    if randomize:
        sd[name + ".weight"] = rng.uniform(0.5, 1.5, c).astype(F32)
        sd[name + ".bias"] = (0.1 * rng.standard_normal(c)).astype(F32)
    else:
        sd[name + ".weight"] = np.ones(c, F32)
        sd[name + ".bias"] = np.zeros(c, F32)


def alexnet_state(seed=1, num_classes=1000):
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    _conv(sd, rng, "features.0", 3, 64, 11, True)
    _conv(sd, rng, "features.3", 64, 192, 5, True)
    _conv(sd, rng, "features.6", 192, 384, 3, True)
    _conv(sd, rng, "features.8", 384, 256, 3, True)
    _conv(sd, rng, "features.10", 256, 256, 3, True)
    _linear(sd, rng, "classifier.1", 256 * 6 * 6, 4096)
    _linear(sd, rng, "classifier.4", 4096, 4096)
    _linear(sd, rng, "classifier.6", 4096, num_classes)
    return sd


VGG_PLANS = {
    "A": (64, "M", 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"),
    "B": (64, 64, "M", 128, 128, "M", 256, 256, "M", 512, 512, "M", 512, 512, "M"),
    "D": (64, 64, "M", 128, 128, "M", 256, 256, 256, "M", 512, 512, 512, "M", 512, 512, 512, "M"),
    "E": (64, 64, "M", 128, 128, "M", 256, 256, 256, 256, "M", 512, 512, 512, 512, "M", 512, 512, 512, 512, "M"),
}


def vgg_state(seed=1, plan="A", batch_norm=False, num_classes=1000):
    """torchvision VGG state_dict layout: `features.{i}` indexes the Sequential [conv, (bn), relu, ..., maxpool, ...];
    classifier Linear layers at 0 / 3 / 6."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    plan = VGG_PLANS[plan] if isinstance(plan, str) else tuple(plan)
    i, cin = 0, 3
    for v in plan:
        if v == "M":
            i += 1
            continue
        _conv(sd, rng, f"features.{i}", cin, v, 3, True)
        i += 1
        if batch_norm:
            _bn(sd, rng, f"features.{i}", v)
            i += 1
        i += 1                      # relu
        cin = v
    _linear(sd, rng, "classifier.0", cin * 7 * 7, 4096)
    _linear(sd, rng, "classifier.3", 4096, 4096)
    _linear(sd, rng, "classifier.6", 4096, num_classes)
    return sd


def resnet_state(seed=1, block="bottleneck", layers=(3, 4, 6, 3), num_classes=1000,
                 width_per_group=64, groups=1, stem=64):
    """torchvision ResNet registration order (mirrored by resnet.py:101-110,171-184)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    exp = 4 if block == "bottleneck" else 1
    _conv(sd, rng, "conv1", 3, stem, 7, False)
    _bn(sd, rng, "bn1", stem)
    inplanes = stem
    for li, (planes, nblk) in enumerate(zip((64, 128, 256, 512), layers)):
        stride = 1 if li == 0 else 2
        for bi in range(nblk):
            p = f"layer{li + 1}.{bi}"
            s = stride if bi == 0 else 1
            if block == "bottleneck":
                width = int(planes * (width_per_group / 64.0)) * groups
                _conv(sd, rng, p + ".conv1", inplanes, width, 1, False)
                _bn(sd, rng, p + ".bn1", width)
                _conv(sd, rng, p + ".conv2", width, width, 3, False, groups)
                _bn(sd, rng, p + ".bn2", width)
                _conv(sd, rng, p + ".conv3", width, planes * 4, 1, False)
                _bn(sd, rng, p + ".bn3", planes * 4)
            else:
                _conv(sd, rng, p + ".conv1", inplanes, planes, 3, False)
                _bn(sd, rng, p + ".bn1", planes)
                _conv(sd, rng, p + ".conv2", planes, planes, 3, False)
                _bn(sd, rng, p + ".bn2", planes)
            if bi == 0 and (s != 1 or inplanes != planes * exp):
                _conv(sd, rng, p + ".downsample.0", inplanes, planes * exp, 1, False)
                _bn(sd, rng, p + ".downsample.1", planes * exp)
            inplanes = planes * exp
    _linear(sd, rng, "fc", 512 * exp, num_classes)
    return sd


MBV2_SETTING = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 3, 2), (6, 64, 4, 2), (6, 96, 3, 1), (6, 160, 3, 2), (6, 320, 1, 1))


def mobilenet_v2_state(seed=1, num_classes=1000, setting=MBV2_SETTING, stem=32, last=1280):
    """torchvision mobilenet_v2 state_dict order: features.0 (conv, bn), features.i.conv.{0,1,2,3} per inverted-residual block
    ([expand conv+bn,] depthwise conv+bn, project conv, bn), features.18 (conv, bn), classifier.1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    _conv(sd, rng, "features.0.0", 3, stem, 3, False)
    _bn(sd, rng, "features.0.1", stem)
    cin, i = stem, 1
    for t, c, n, s in setting:
        for _ in range(n):
            hidden = int(round(cin * t))
            p, j = f"features.{i}.conv", 0
            if t != 1:
                _conv(sd, rng, f"{p}.{j}.0", cin, hidden, 1, False)
                _bn(sd, rng, f"{p}.{j}.1", hidden)
                j += 1
            _conv(sd, rng, f"{p}.{j}.0", hidden, hidden, 3, False, groups=hidden)
            _bn(sd, rng, f"{p}.{j}.1", hidden)
            _conv(sd, rng, f"{p}.{j + 1}", hidden, c, 1, False)
            _bn(sd, rng, f"{p}.{j + 2}", c)
            cin = c
            i += 1
    _conv(sd, rng, f"features.{i}.0", cin, last, 1, False)
    _bn(sd, rng, f"features.{i}.1", last)
    _linear(sd, rng, "classifier.1", last, num_classes)
    return sd


def _md(v, d=8):
    n = max(d, int(v + d / 2) // d * d)
    return n + d if n < 0.9 * v else n


def mobilenet_v3_conf(arch="large", dilated=False, reduced_tail=False):
    """Rows (in, kernel, expanded, out, use_se, "RE"|"HS", stride, dilation) of the two published tables (mobilenetv3.py:266-338),
    channels already rounded to multiples of 8; plus the width of the classifier's hidden layer."""
    dv, dl = (2 if reduced_tail else 1), (2 if dilated else 1)
    if arch == "large":
        t = [(16, 3, 16, 16, False, "RE", 1, 1), (16, 3, 64, 24, False, "RE", 2, 1), (24, 3, 72, 24, False, "RE", 1, 1),
             (24, 5, 72, 40, True, "RE", 2, 1), (40, 5, 120, 40, True, "RE", 1, 1), (40, 5, 120, 40, True, "RE", 1, 1),
             (40, 3, 240, 80, False, "HS", 2, 1), (80, 3, 200, 80, False, "HS", 1, 1), (80, 3, 184, 80, False, "HS", 1, 1),
             (80, 3, 184, 80, False, "HS", 1, 1), (80, 3, 480, 112, True, "HS", 1, 1), (112, 3, 672, 112, True, "HS", 1, 1),
             (112, 5, 672, 160 // dv, True, "HS", 2, dl), (160 // dv, 5, 960 // dv, 160 // dv, True, "HS", 1, dl),
             (160 // dv, 5, 960 // dv, 160 // dv, True, "HS", 1, dl)]
        last = 1280 // dv
    else:
        t = [(16, 3, 16, 16, True, "RE", 2, 1), (16, 3, 72, 24, False, "RE", 2, 1), (24, 3, 88, 24, False, "RE", 1, 1),
             (24, 5, 96, 40, True, "HS", 2, 1), (40, 5, 240, 40, True, "HS", 1, 1), (40, 5, 240, 40, True, "HS", 1, 1),
             (40, 5, 120, 48, True, "HS", 1, 1), (48, 5, 144, 48, True, "HS", 1, 1), (48, 5, 288, 96 // dv, True, "HS", 2, dl),
             (96 // dv, 5, 576 // dv, 96 // dv, True, "HS", 1, dl), (96 // dv, 5, 576 // dv, 96 // dv, True, "HS", 1, dl)]
        last = 1024 // dv
    return [(_md(a), k, _md(e), _md(o), se, act, s, d) for a, k, e, o, se, act, s, d in t], _md(last)


def _mbv3_features(sd, rng, conf, prefix):
    _conv(sd, rng, prefix + "0.0", 3, conf[0][0], 3, False)
    _bn(sd, rng, prefix + "0.1", conf[0][0])
    for i, (cin, k, cexp, cout, use_se, a, stride, dil) in enumerate(conf, start=1):
        p, j = f"{prefix}{i}.block", 0
        if cexp != cin:
            _conv(sd, rng, f"{p}.0.0", cin, cexp, 1, False)
            _bn(sd, rng, f"{p}.0.1", cexp)
            j = 1
        _conv(sd, rng, f"{p}.{j}.0", cexp, cexp, k, False, groups=cexp)
        _bn(sd, rng, f"{p}.{j}.1", cexp)
        j += 1
        if use_se:
            sq = _md(cexp // 4)
            _conv(sd, rng, f"{p}.{j}.fc1", cexp, sq, 1, True)
            _conv(sd, rng, f"{p}.{j}.fc2", sq, cexp, 1, True)
            j += 1
        _conv(sd, rng, f"{p}.{j}.0", cexp, cout, 1, False)
        _bn(sd, rng, f"{p}.{j}.1", cout)
    i = len(conf) + 1
    _conv(sd, rng, f"{prefix}{i}.0", conf[-1][3], 6 * conf[-1][3], 1, False)
    _bn(sd, rng, f"{prefix}{i}.1", 6 * conf[-1][3])
    return 6 * conf[-1][3]


def mobilenet_v3_state(seed=1, conf=None, last=1280, num_classes=1000):
    """torchvision mobilenet_v3 state_dict order: features.{i}[.block.{j}] ..., classifier.0, classifier.3."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    if conf is None:
        conf, last = mobilenet_v3_conf("large")
    width = _mbv3_features(sd, rng, conf, "features.")
    _linear(sd, rng, "classifier.0", width, last)
    _linear(sd, rng, "classifier.3", last, num_classes)
    return sd


def lraspp_state(seed=1, conf=None, taps=(4, 16), num_classes=21, inter=128):
    """torchvision lraspp_mobilenet_v3_large order: backbone.{i}..., classifier.{cbr.0, cbr.1, scale.1, low_classifier, high_classifier}."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    if conf is None:
        conf, _ = mobilenet_v3_conf("large", dilated=True)
    width = _mbv3_features(sd, rng, conf, "backbone.")
    outc = lambda i: conf[0][0] if i == 0 else (width if i == len(conf) + 1 else conf[i - 1][3])
    low_c, high_c = outc(taps[0]), outc(taps[1])
    _conv(sd, rng, "classifier.cbr.0", high_c, inter, 1, False)
    _bn(sd, rng, "classifier.cbr.1", inter)
    _conv(sd, rng, "classifier.scale.1", high_c, inter, 1, False)
    _conv(sd, rng, "classifier.low_classifier", low_c, num_classes, 1, True)
    _conv(sd, rng, "classifier.high_classifier", inter, num_classes, 1, True)
    return sd


def efficientnet_stages(arch="b0"):
    """Stage rows (fused, expand, kernel, stride, in, out, layers) AFTER width / depth scaling, the width of the last 1x1 conv and the
    BatchNorm eps (efficientnet.py:404-715)."""
    import math
    b = {"b0": (1.0, 1.0), "b1": (1.0, 1.1), "b2": (1.1, 1.2), "b3": (1.2, 1.4), "b4": (1.4, 1.8), "b5": (1.6, 2.2), "b6": (1.8, 2.6),
         "b7": (2.0, 3.1)}
    if arch in b:
        wm, dm = b[arch]
        t = ((1, 3, 1, 32, 16, 1), (6, 3, 2, 16, 24, 2), (6, 5, 2, 24, 40, 2), (6, 3, 2, 40, 80, 3), (6, 5, 1, 80, 112, 3),
             (6, 5, 2, 112, 192, 4), (6, 3, 1, 192, 320, 1))
        rows = [(0, e, k, s, _md(i * wm), _md(o * wm), int(math.ceil(n * dm))) for e, k, s, i, o, n in t]
        return rows, 4 * rows[-1][5], (1e-3 if arch in ("b5", "b6", "b7") else 1e-5)
    v2 = {"v2_s": ((1, 1, 3, 1, 24, 24, 2), (1, 4, 3, 2, 24, 48, 4), (1, 4, 3, 2, 48, 64, 4), (0, 4, 3, 2, 64, 128, 6),
                   (0, 6, 3, 1, 128, 160, 9), (0, 6, 3, 2, 160, 256, 15))}
    return list(v2[arch]), 1280, 1e-3


def efficientnet_state(seed=1, stages=None, last=None, num_classes=1000):
    """torchvision efficientnet state_dict order: features.0 (conv, bn), features.{s}.{b}.block.{j}..., features.{S+1}, classifier.1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    if stages is None:
        stages, last, _ = efficientnet_stages("b0")
    _conv(sd, rng, "features.0.0", 3, stages[0][4], 3, False)
    _bn(sd, rng, "features.0.1", stages[0][4])
    for si, (fused, e, k, s, cin, cout, n) in enumerate(stages, start=1):
        for b in range(n):
            ci = cin if b == 0 else cout
            p = f"features.{si}.{b}.block"
            cexp = _md(ci * e)
            if fused:
                if cexp != ci:
                    _conv(sd, rng, p + ".0.0", ci, cexp, k, False)
                    _bn(sd, rng, p + ".0.1", cexp)
                    _conv(sd, rng, p + ".1.0", cexp, cout, 1, False)
                    _bn(sd, rng, p + ".1.1", cout)
                else:
                    _conv(sd, rng, p + ".0.0", ci, cout, k, False)
                    _bn(sd, rng, p + ".0.1", cout)
                continue
            j = 0
            if cexp != ci:
                _conv(sd, rng, f"{p}.0.0", ci, cexp, 1, False)
                _bn(sd, rng, f"{p}.0.1", cexp)
                j = 1
            _conv(sd, rng, f"{p}.{j}.0", cexp, cexp, k, False, groups=cexp)
            _bn(sd, rng, f"{p}.{j}.1", cexp)
            sq = max(1, ci // 4)
            _conv(sd, rng, f"{p}.{j + 1}.fc1", cexp, sq, 1, True)
            _conv(sd, rng, f"{p}.{j + 1}.fc2", sq, cexp, 1, True)
            _conv(sd, rng, f"{p}.{j + 2}.0", cexp, cout, 1, False)
            _bn(sd, rng, f"{p}.{j + 2}.1", cout)
    i = len(stages) + 1
    _conv(sd, rng, f"features.{i}.0", stages[-1][5], last, 1, False)
    _bn(sd, rng, f"features.{i}.1", last)
    _linear(sd, rng, "classifier.1", last, num_classes)
    return sd


def regnet_state(seed=1, widths=(48, 104, 208, 440), depths=(1, 3, 6, 6), group_widths=(8, 8, 8, 8), se_ratio=0.25, stem=32,
                 num_classes=1000):
    """torchvision regnet state_dict ORDER (names follow this package's Sequential indices): stem conv + bn; per block [proj conv + bn,]
    a (1x1) conv + bn, b (3x3 grouped) conv + bn, [se fc1, fc2,] c (1x1) conv + bn; fc.  Bottleneck multiplier 1."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    _conv(sd, rng, "stem.0", 3, stem, 3, False)
    _bn(sd, rng, "stem.1", stem)
    win = stem
    for si, (w, d, gw) in enumerate(zip(widths, depths, group_widths)):
        for b in range(d):
            p = f"trunk_output.{si}.{b}"
            stride = 2 if b == 0 else 1
            if win != w or stride != 1:
                _conv(sd, rng, p + ".proj.0", win, w, 1, False)
                _bn(sd, rng, p + ".proj.1", w)
            _conv(sd, rng, p + ".f.0.0", win, w, 1, False)
            _bn(sd, rng, p + ".f.0.1", w)
            _conv(sd, rng, p + ".f.1.0", w, w, 3, False, groups=w // gw)
            _bn(sd, rng, p + ".f.1.1", w)
            j = 2
            if se_ratio:
                sq = int(round(se_ratio * win))
                _conv(sd, rng, p + ".f.2.fc1", w, sq, 1, True)
                _conv(sd, rng, p + ".f.2.fc2", sq, w, 1, True)
                j = 3
            _conv(sd, rng, f"{p}.f.{j}.0", w, w, 1, False)
            _bn(sd, rng, f"{p}.f.{j}.1", w)
            win = w
    _linear(sd, rng, "fc", win, num_classes)
    return sd


def segmentation_state(seed=1, kind="fcn", layers=(3, 4, 6, 3), num_classes=21, aux=True):
    """torchvision fcn_resnet50 / deeplabv3_resnet50 state_dict order: backbone (ResNet without fc), classifier, aux_classifier."""
    rng = np.random.Generator(np.random.PCG64(seed + 100))
    sd = OrderedDict()
    for k, v in resnet_state(seed, "bottleneck", layers, 10).items():
        if not k.startswith("fc."):
            sd["backbone." + k] = v

    def fcn_head(p, cin):
        _conv(sd, rng, p + ".0", cin, cin // 4, 3, False)
        _bn(sd, rng, p + ".1", cin // 4)
        _conv(sd, rng, p + ".4", cin // 4, num_classes, 1, True)

    if kind == "fcn":
        fcn_head("classifier", 2048)
    else:
        a = "classifier.0"
        _conv(sd, rng, a + ".convs.0.0", 2048, 256, 1, False)
        _bn(sd, rng, a + ".convs.0.1", 256)
        for i in range(3):
            _conv(sd, rng, f"{a}.convs.{i + 1}.0", 2048, 256, 3, False)
            _bn(sd, rng, f"{a}.convs.{i + 1}.1", 256)
        _conv(sd, rng, a + ".convs.4.1", 2048, 256, 1, False)
        _bn(sd, rng, a + ".convs.4.2", 256)
        _conv(sd, rng, a + ".project.0", 5 * 256, 256, 1, False)
        _bn(sd, rng, a + ".project.1", 256)
        _conv(sd, rng, "classifier.1", 256, 256, 3, False)
        _bn(sd, rng, "classifier.2", 256)
        _conv(sd, rng, "classifier.4", 256, num_classes, 1, True)
    if aux:
        fcn_head("aux_classifier", 1024)
    return sd


def _trunc_normal(rng, shape):
    x = rng.standard_normal(shape)
    bad = np.abs(x) > 2
    while bad.any():
        x[bad] = rng.standard_normal(int(bad.sum()))
        bad = np.abs(x) > 2
    return x.astype(F32)


def vit_state(seed=1, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12,
              mlp_ratio=4, num_classes=1000, in_chans=3, qkv_bias=True, randomize_ln=True):
    """DINO/timm ViT state_dict order (vit.py:163-171 field order)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    n = (img_size // patch_size) ** 2
    sd["cls_token"] = _trunc_normal(rng, (1, 1, embed_dim))         # vit.py:229-231 (unit scale)
    sd["pos_embed"] = _trunc_normal(rng, (1, n + 1, embed_dim))     # vit.py:232-234
    _conv(sd, rng, "patch_embed.proj", in_chans, embed_dim, patch_size, True)
    hidden = int(embed_dim * mlp_ratio)
    for i in range(depth):
        p = f"blocks.{i}"
        _ln(sd, rng, p + ".norm1", embed_dim, randomize_ln)
        _linear(sd, rng, p + ".attn.qkv", embed_dim, 3 * embed_dim, qkv_bias)
        _linear(sd, rng, p + ".attn.proj", embed_dim, embed_dim)
        _ln(sd, rng, p + ".norm2", embed_dim, randomize_ln)
        _linear(sd, rng, p + ".mlp.fc1", embed_dim, hidden)
        _linear(sd, rng, p + ".mlp.fc2", hidden, embed_dim)
    _ln(sd, rng, "norm", embed_dim, randomize_ln)
    if num_classes:
        _linear(sd, rng, "fc", embed_dim, num_classes)
    return sd


def swin_relative_position_index(ws):
    """torchvision's index (what pretrained checkpoints carry and overwrite the reference's
    own buggy init with -- SURVEY Appendix C-1 / D)."""
    ch, cw = np.arange(ws[0]), np.arange(ws[1])
    coords = np.stack(np.meshgrid(ch, cw, indexing="ij")).reshape(2, -1)
    rel = (coords[:, :, None] - coords[:, None, :]).transpose(1, 2, 0).copy()
    rel[:, :, 0] += ws[0] - 1
    rel[:, :, 1] += ws[1] - 1
    rel[:, :, 0] *= 2 * ws[1] - 1
    return rel.sum(-1).reshape(-1).astype(np.int64)


def swin_state(seed=1, patch_size=(4, 4), embed_dim=96, depths=(2, 2, 6, 2), num_heads=(3, 6, 12, 24),
               window_size=(7, 7), mlp_ratio=4.0, num_classes=1000, randomize_ln=True):
    """torchvision swin_t registration order (swin.py:260-268,526-530,47-48)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    sd = OrderedDict()
    _conv(sd, rng, "features.0.0", 3, embed_dim, tuple(patch_size), True)
    _ln(sd, rng, "features.0.2", embed_dim, randomize_ln)
    fi = 1
    for si, depth in enumerate(depths):
        dim = embed_dim * 2 ** si
        for bi in range(depth):
            p = f"features.{fi}.{bi}"
            _ln(sd, rng, p + ".norm1", dim, randomize_ln)
            sd[p + ".attn.relative_position_bias_table"] = (
                0.02 * rng.standard_normal(((2 * window_size[0] - 1) * (2 * window_size[1] - 1), num_heads[si]))
            ).astype(F32)
            sd[p + ".attn.relative_position_index"] = swin_relative_position_index(window_size)
            _linear(sd, rng, p + ".attn.qkv", dim, 3 * dim)
            _linear(sd, rng, p + ".attn.proj", dim, dim)
            _ln(sd, rng, p + ".norm2", dim, randomize_ln)
            _linear(sd, rng, p + ".mlp.0", dim, int(dim * mlp_ratio))
            _linear(sd, rng, p + ".mlp.3", int(dim * mlp_ratio), dim)
        fi += 1
        if si < len(depths) - 1:
            p = f"features.{fi}"
            _linear(sd, rng, p + ".reduction", 4 * dim, 2 * dim, bias=False)
            _ln(sd, rng, p + ".norm", 4 * dim, randomize_ln)
            fi += 1
    nf = embed_dim * 2 ** (len(depths) - 1)
    _ln(sd, rng, "norm", nf, randomize_ln)
    _linear(sd, rng, "head", nf, num_classes)
    return sd


def save_pth(sd, path):
    """Write the dict as a real torch checkpoint so `load_torch_weights` exercises torch.load."""
    import torch
    torch.save(OrderedDict((k, torch.from_numpy(np.ascontiguousarray(v))) for k, v in sd.items()), path)


def synthetic_images(batch, size=224, seed=0, chans=3):
    """U[0,1) fp32 NCHW images, PCG64(seed)  (reference README.md:45 uses jr.uniform)."""
    rng = np.random.Generator(np.random.PCG64(seed))
    return rng.random((batch, chans, size, size), dtype=F32)
