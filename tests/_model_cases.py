"""Model-level GPU parity cases: product (eqxvision_amd, HIP through the C ABI) vs CPU oracle on
the same synthetic checkpoint (loaded through `load_torch_weights`) and the same seeded images.
Tolerance (north star): logits within 1e-2 (bf16) / 1e-3 (fp32) ABSOLUTE; `err_scaled` = err / max(1, max|ref|) is reported too."""
from __future__ import annotations

import os
import tempfile

import numpy as np
import torch

from oracle import models as OM
from oracle import np_ops as O
from oracle import state as S
from oracle import torch_ref as TR


def _cmp(got, ref, tol, extra=None, scaled=False):
    got = np.asarray(got, np.float64)
    ref = np.asarray(ref, np.float64)
    if got.shape != ref.shape:
        return {"ok": False, "err": f"shape {got.shape} vs {ref.shape}"}
    if not np.isfinite(got).all():
        return {"ok": False, "err": "non-finite output"}
    d = float(np.abs(got - ref).max())
    # logits: ABSOLUTE, "within 1e-3 fp32 / 1e-2 bf16" (BASELINE.json north_star); feature maps (scaled=True): relative to max|ref|
    lim = tol * max(1.0, float(np.abs(ref).max())) if scaled else tol
    out = {"ok": d <= lim, "err": d, "lim": lim, "refmax": float(np.abs(ref).max()),
           "err_scaled": d / max(1.0, float(np.abs(ref).max())),
           "argmax_match": bool((got.reshape(got.shape[0], -1).argmax(-1) == ref.reshape(ref.shape[0], -1).argmax(-1)).all())}
    if extra:
        out.update(extra)
    return out


def _load(factory, sd, **kw):
    import eqxvision_amd as eqv
    with tempfile.TemporaryDirectory() as d:
        p = os.path.join(d, "w.pth")
        S.save_pth(sd, p)
        net = factory(torch_weights=p, **kw)
    return eqv.tree_inference(net, True)


def _keys(B):
    import eqxvision_amd as eqv
    return eqv.random.split(eqv.random.PRNGKey(0), B)


def _run(net, x, dtype, jit=False):
    import eqxvision_amd as eqv
    with eqv.precision(dtype):
        if jit:
            fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))
            return fwd, fwd(net, x, _keys(x.shape[0]))
        out = eqv.vmap(net, axis_name="batch")(x, key=_keys(x.shape[0]))
    torch.cuda.synchronize()
    return out


def resnet_case(block, layers, size, B, classes=10, dtype="bf16", full_ref="numpy", groups=1, width_per_group=64):
    def run():
        import eqxvision_amd as eqv
        sd = S.resnet_state(1, block, layers, classes, width_per_group=width_per_group, groups=groups)
        x = S.synthetic_images(B, size, seed=0)
        fac = {("bottleneck", (3, 4, 6, 3)): eqv.models.resnet50, ("basic", (2, 2, 2, 2)): eqv.models.resnet18}.get(
            (block, tuple(layers))) if groups == 1 else None
        if groups > 1 and tuple(layers) == (3, 4, 6, 3) and width_per_group == 4:
            fac = eqv.models.resnext50_32x4d
        if fac is None and groups > 1:
            blk = eqv.models.classification.resnet._ResNetBottleneck
            fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(
                blk, list(layers), torch_weights, groups=groups, width_per_group=width_per_group, **kw)
        if fac is None:
            blk = eqv.models.classification.resnet._ResNetBottleneck if block == "bottleneck" else \
                eqv.models.classification.resnet._ResNetBasicBlock
            fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, list(layers), torch_weights, **kw)
        net = _load(fac, sd, num_classes=classes)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.resnet_forward(sd, x, block, layers, groups).numpy()
            emu = None
        else:
            ref = O.vmap(lambda im: OM.resnet_forward(sd, im, block, layers, groups=groups))(x)
            emu = O.vmap(lambda im: OM.resnet_forward(sd, im, block, layers, bf16=True, groups=groups))(x) if dtype == "bf16" else None
        extra = {}
        if emu is not None:
            extra["err_vs_bf16_emulation"] = float(np.abs(got - emu).max())
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3, extra)
    return run


def resnet_train_case(B=8, size=64, classes=10, jit=False):
    """resnet18 with every BatchNorm in TRAINING mode (reference resnet.py:132-136 / 252 / 301 before tree_inference; the
    semantics of eqx.experimental.BatchNorm's training branch): two steps -- batch statistics, running
    statistics by EMA, normalisation with the updated running statistics -- against oracle.models.resnet_forward_train, the
    running statistics of the first and last BatchNorm against the oracle's, then an INFERENCE forward of the same modules
    (shared state) against the oracle's inference forward with the updated statistics."""
    def run():
        import eqxvision_amd as eqv
        block, layers = "basic", (2, 2, 2, 2)
        sd = S.resnet_state(1, block, layers, classes)
        net_inf = _load(eqv.models.resnet18, sd, num_classes=classes)
        net = eqv.tree_inference(net_inf, False)
        running = {k[:-len(".running_mean")]: (sd[k], sd[k[:-len("mean")] + "var"]) for k in sd if k.endswith(".running_mean")}
        fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k)) if jit else \
            (lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))
        errs, info = [], {}
        with eqv.precision("bf16"):
            for step in range(4 if jit else 2):       # under filter_jit: record, capture + launch, graph launch, graph launch
                x = S.synthetic_images(B, size, seed=step)
                got = fwd(net, x, _keys(B))
                torch.cuda.synchronize()
                ref = OM.resnet_forward_train(sd, x, running, block, layers, bf16=True)
                info[f"refmax_step{step}"] = float(np.abs(ref).max())
                errs.append(float(np.abs(got.cpu().numpy() - ref).max()) / max(1.0, float(np.abs(ref).max())))
            for name, mod in (("bn1", net.bn1), ("layer4.1.bn2", net.layer4.layers[1].bn2)):
                m, v = mod.state_index.value
                info[f"{name}_running_err"] = max(float(np.abs(np.asarray(m) - running[name][0]).max()),
                                                  float(np.abs(np.asarray(v) - running[name][1]).max() /
                                                        max(1.0, float(np.abs(running[name][1]).max()))))
            sd2 = dict(sd)
            for name, (m, v) in running.items():
                sd2[name + ".running_mean"], sd2[name + ".running_var"] = m, v
            x = S.synthetic_images(B, size, seed=7)
            got = eqv.vmap(net_inf, axis_name="batch")(x, key=_keys(B)).cpu().numpy()
            ref = O.vmap(lambda im: OM.resnet_forward(sd2, im, block, layers, bf16=True))(x)
            errs.append(float(np.abs(got - ref).max()))
        # bf16 against the bf16-emulating oracle; training steps relative to max|logit| (un-normalised activations grow: the
        # reference normalises with the RUNNING statistics even in training mode)
        info.update({"err": max(errs), "lim": 1e-2, "train_step_errs_scaled": errs[:-1], "inference_after_err": errs[-1]})
        info["ok"] = max(errs) <= 1e-2 and all(v < 1e-2 for k, v in info.items() if k.endswith("_running_err"))
        if jit:           # the steps after the first were replays of ONE recording (a hipGraph from the second on), not re-traces
            ent = [c for c in fwd._entries() if c.graph is not None]
            info["graph_replays"] = max([c.replays for c in ent], default=0)
            info["ok"] = info["ok"] and len(fwd._entries()) == 1 and info["graph_replays"] >= 3
        return info
    return run


def stochastic_layers_case():
    """Dropout / DropPath outside inference mode: the reference's own checks (tests/test_layers.py:36-68:
    shape kept, some zeros; none after tree_inference) and, beyond them, the SAME elements dropped as eqx.nn.Dropout /
    DropPath would drop for these keys (oracle.np_ops.dropout / drop_path on JAX's bit stream) -- outputs equal bit for bit."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import layers, nn
        jr = eqv.random
        info, ok = {}, True
        rng = np.random.Generator(np.random.PCG64(5))
        keys = jr.split(jr.PRNGKey(11), 10)
        x = rng.uniform(1, 10, (10, 20)).astype(np.float32)                              # tests/test_layers.py:43
        bfr = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
        for name, mod, ref in (("drop_path_global", layers.DropPath(0.5, mode="global"), lambda v, k: O.drop_path(v, 0.5, "global", k)),
                               ("drop_path_local", layers.DropPath(0.5, mode="local"), lambda v, k: O.drop_path(v, 0.5, "local", k)),
                               ("dropout_vec", nn.Dropout(0.3), lambda v, k: O.dropout(v, 0.3, k))):
            with eqv.precision("fp32"):
                got = eqv.filter_jit(lambda m, a, k: eqv.vmap(m)(a, key=k))(mod, x, keys).cpu().numpy()
                idn = eqv.vmap(eqv.tree_inference(mod, True))(x, key=keys).cpu().numpy()
            want = np.stack([ref(x[i], keys[i]) for i in range(10)])
            good = got.shape == (10, 20) and bool((got == 0).any()) and bool((idn != 0).all()) and np.array_equal(got, want)
            info[name] = {"bit_identical": bool(np.array_equal(got, want)), "zeros": int((got == 0).sum())}
            ok = ok and good
        # feature maps (C,H,W) in bf16: the mask follows the LOGICAL (C,H,W) order although the device layout is NHWC
        xm = bfr(rng.standard_normal((6, 24, 5, 7)))
        km = jr.split(jr.PRNGKey(12), 6)
        with eqv.precision("bf16"):
            got = eqv.vmap(nn.Dropout(0.4))(xm, key=km).cpu().numpy()
            gl = eqv.vmap(layers.DropPath(0.5, mode="local"))(xm, key=km).cpu().numpy()
        want = np.stack([bfr(O.dropout(xm[i], 0.4, km[i])) for i in range(6)])
        wl = np.stack([bfr(O.drop_path(xm[i], 0.5, "local", km[i])) for i in range(6)])
        info["dropout_map_bf16"] = {"bit_identical": bool(np.array_equal(got, want)), "kept": float((got != 0).mean())}
        info["drop_path_local_map_bf16"] = {"bit_identical": bool(np.array_equal(gl, wl))}
        # token matrices (N,D), odd element count (the padded counter)
        xs = rng.standard_normal((3, 7, 9)).astype(np.float32)
        ksq = jr.split(jr.PRNGKey(13), 3)
        with eqv.precision("fp32"):
            got = eqv.vmap(nn.Dropout(0.5))(xs, key=ksq).cpu().numpy()
        want = np.stack([O.dropout(xs[i], 0.5, ksq[i]) for i in range(3)])
        info["dropout_seq_odd"] = {"bit_identical": bool(np.array_equal(got, want))}
        ok = ok and all(v["bit_identical"] for v in info.values())
        info.update(ok=ok, err=0.0 if ok else 1.0, lim=0.0)
        return info
    return run


def alexnet_train_case(B=2):
    """AlexNet outside inference mode: the classifier's Dropout(0.5) pair is live (alexnet.py:63-68); keys split as the reference
    splits them (alexnet.py:80-84 + nn.Sequential) -> the same activations dropped as in the reference -> logits within the
    bf16 tolerance of the oracle that applies the same masks."""
    def run():
        import eqxvision_amd as eqv
        sd = S.alexnet_state(1, 1000)
        x = S.synthetic_images(B, 224, seed=0)
        net = eqv.tree_inference(_load(eqv.models.alexnet, sd), False)
        keys = eqv.random.split(eqv.random.PRNGKey(21), B)
        with eqv.precision("bf16"):
            fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))
            got = fwd(net, x, keys).cpu().numpy()
            again = fwd(net, x, eqv.random.split(eqv.random.PRNGKey(22), B)).cpu().numpy()
        ref = np.stack([OM.alexnet_forward(sd, x[i], bf16=True, key=keys[i]) for i in range(B)])
        inf = np.stack([OM.alexnet_forward(sd, x[i], bf16=True) for i in range(B)])
        out = _cmp(got, ref, 1e-2, scaled=True)
        out["differs_from_inference"] = float(np.abs(got - inf).max())
        out["other_keys_differ"] = float(np.abs(got - again).max())
        out["ok"] = out["ok"] and out["differs_from_inference"] > 10 * out["err"] and out["other_keys_differ"] > 10 * out["err"]
        return out
    return run


def alexnet_case(B, dtype="bf16", features_only=False):
    def run():
        import eqxvision_amd as eqv
        sd = S.alexnet_state(1, 1000)
        x = S.synthetic_images(B, 224, seed=0)
        net = _load(eqv.models.alexnet, sd)
        if features_only:      # what the reference's own test pins (tests/test_models/test_alexnet.py:18-25)
            with eqv.precision(dtype):
                got = eqv.vmap(net.features, axis_name="batch")(x, key=_keys(B)).cpu().numpy()
            ref = TR.alexnet_features(sd, x).numpy()
        else:
            got = _run(net, x, dtype).cpu().numpy()
            ref = TR.alexnet_forward(sd, x).numpy()
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3, scaled=features_only)
    return run


def mobilenet_v2_case(setting, size, B, classes=10, last=1280, dtype="bf16", full_ref="numpy"):
    """MobileNetV2 (reference models/classification/mobilenetv2.py): depthwise + odd-width pointwise layers, residual in the GEMM
    epilogue; `setting` None = the published architecture."""
    def run():
        import eqxvision_amd as eqv
        st = tuple(tuple(r) for r in setting) if setting is not None else S.MBV2_SETTING
        sd = S.mobilenet_v2_state(1, classes, st, last=last)
        x = S.synthetic_images(B, size, seed=0)
        kw = {} if setting is None else {"inverted_residual_setting": [list(r) for r in st]}
        if setting is None:
            fac = eqv.models.mobilenet_v2
        else:      # a reduced stack: same constructor, the last 1x1 conv narrowed through width rounding is not available -> patch it
            def fac(torch_weights=None, **k):
                net = eqv.models.MobileNetV2(**k)
                if last != 1280:
                    from eqxvision_amd.layers import ConvNormActivation
                    from eqxvision_amd._module import tree_at
                    cin = net.features.layers[-1].layers[0].in_channels
                    net = tree_at(lambda m: (m.features.layers[-1], m.classifier.layers[-1]), net,
                                  (ConvNormActivation(cin, last, kernel_size=1, key=eqv.random.PRNGKey(5)),
                                   eqv.nn.Linear(last, k.get("num_classes", 1000), key=eqv.random.PRNGKey(6))))
                return eqv.utils.load_torch_weights(net, torch_weights)
        net = _load(fac, sd, num_classes=classes, **kw)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.mobilenet_v2_forward(sd, x, st).numpy()
        else:
            ref = np.stack([OM.mobilenet_v2_forward(sd, im, st) for im in x])
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3)
    return run


def mobilenet_v3_case(arch, size, B, classes=10, dtype="bf16", full_ref="numpy", rows=None):
    """MobileNetV3 (reference mobilenetv3.py): SE blocks, hard_swish / hard_sigmoid, 5x5 depthwise; `rows` truncates the table."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd.models.classification import mobilenetv3 as M3
        conf, last = S.mobilenet_v3_conf(arch)
        if rows:
            conf, last = conf[:rows], 64
        sd = S.mobilenet_v3_state(1, conf, last, classes)
        x = S.synthetic_images(B, size, seed=0)
        if rows:
            setting, _ = M3._mobilenet_v3_conf(f"mobilenet_v3_{arch}")
            fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(M3.MobileNetV3(setting[:rows], last, **kw), torch_weights)
        else:
            fac = eqv.models.mobilenet_v3_large if arch == "large" else eqv.models.mobilenet_v3_small
        net = _load(fac, sd, num_classes=classes)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.mobilenet_v3_forward(sd, x, conf).numpy()
        else:
            ref = np.stack([OM.mobilenet_v3_forward(sd, im, conf) for im in x])
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3)
    return run


def lraspp_case(size, B, classes=21, dtype="bf16", full_ref="torch", jit=False):
    """lraspp_mobilenet_v3_large (reference lraspp.py, tests/test_models/test_lraspp.py): dilated backbone, taps [4, 16]."""
    def run():
        import eqxvision_amd as eqv
        conf, _ = S.mobilenet_v3_conf("large", dilated=True)
        sd = S.lraspp_state(1, conf, (4, 16), classes)
        x = S.synthetic_images(B, size, seed=0)
        net = _load(lambda torch_weights=None, **kw: eqv.models.lraspp_mobilenet_v3_large(torch_weights=torch_weights, **kw), sd,
                    num_classes=classes)
        if jit:
            fwd, got = _run(net, x, dtype, jit=True)
            with eqv.precision(dtype):
                for _ in range(3):
                    got = fwd(net, x, _keys(B))
            torch.cuda.synchronize()
        else:
            got = _run(net, x, dtype)
        none, out = got
        if full_ref == "torch":
            ref = TR.lraspp_forward(sd, x, conf).numpy()
        else:
            ref = np.stack([OM.lraspp_forward(sd, im, conf) for im in x])
        info = _cmp(out.cpu().numpy(), ref, 1e-2 if dtype == "bf16" else 1e-3)
        info["ok"] = info["ok"] and none is None and tuple(out.shape) == (B, classes, size, size)
        return info
    return run


def efficientnet_case(arch, size, B, classes=10, dtype="bf16", full_ref="numpy", stages=None, last=64):
    """EfficientNet / EfficientNetV2 (reference efficientnet.py): MBConv with SE + SiLU, FusedMBConv; `stages` = a reduced table."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd.models.classification import efficientnet as E
        if stages is None:
            st, lastc, eps = S.efficientnet_stages(arch)
            fac = getattr(eqv.models, f"efficientnet_{arch}")
        else:
            st, lastc, eps = list(stages), last, 1e-5
            cfgs = [(E._FusedMBConvConfig if f else E._MBConvConfig)(e, k, s_, i, o, n) for f, e, k, s_, i, o, n in st]
            fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(E.EfficientNet(cfgs, 0.2, last_channel=lastc, **kw),
                                                                                torch_weights)
        sd = S.efficientnet_state(1, st, lastc, classes)
        x = S.synthetic_images(B, size, seed=0)
        net = _load(fac, sd, num_classes=classes)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.efficientnet_forward(sd, x, st, eps).numpy()
        else:
            ref = np.stack([OM.efficientnet_forward(sd, im, st, eps) for im in x])
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3)
    return run


def regnet_case(name, size, B, classes=10, dtype="bf16", full_ref="numpy", custom=None):
    """RegNet (reference regnet.py): grouped 3x3 convolutions through per-tile input windows, SE for the Y variants, shortcut + relu in
    the last GEMM's epilogue.  `custom` = (widths, depths, group_widths, se_ratio) for a reduced net."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd.models.classification import regnet as R
        if custom is None:
            bp = R.BlockParams.from_init_params(**R._CONFIGS[name])
            fac = getattr(eqv.models, name)
        else:
            w, d, g, se = custom
            bp = R.BlockParams(list(d), list(w), list(g), [1.0] * len(w), [2] * len(w), se)
            fac = lambda torch_weights=None, **kw: R._regnet("custom", bp, torch_weights, **kw)
        sd = S.regnet_state(1, bp.widths, bp.depths, bp.group_widths, bp.se_ratio, 32, classes)
        x = S.synthetic_images(B, size, seed=0)
        net = _load(fac, sd, num_classes=classes)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.regnet_forward(sd, x, bp.widths, bp.depths, bp.group_widths, bp.se_ratio).numpy()
        else:
            ref = np.stack([OM.regnet_forward(sd, im, bp.widths, bp.depths, bp.group_widths, bp.se_ratio) for im in x])
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3)
    return run


def vgg_case(plan, batch_norm, size, B, classes=10, dtype="bf16", full_ref="numpy"):
    """VGG (reference models/classification/vgg.py) incl. its single-relu classifier; `plan` = a torchvision letter or a list."""
    def run():
        import eqxvision_amd as eqv
        cfg = list(S.VGG_PLANS[plan]) if isinstance(plan, str) else list(plan)
        sd = S.vgg_state(1, cfg, batch_norm, classes)
        x = S.synthetic_images(B, size, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VGG(**kw), torch_weights)
        net = _load(fac, sd, cfg=cfg, batch_norm=batch_norm, num_classes=classes)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.vgg_forward(sd, x, cfg, batch_norm).numpy()
            extra = {}
        else:
            ref = O.vmap(lambda im: OM.vgg_forward(sd, im, cfg, batch_norm))(x)
            extra = {}
            if dtype == "bf16":
                emu = O.vmap(lambda im: OM.vgg_forward(sd, im, cfg, batch_norm, bf16=True))(x)
                extra["err_vs_bf16_emulation"] = float(np.abs(got - emu).max())
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3, extra)
    return run


def segmentation_case(kind, layers, size, B, classes=5, dtype="bf16", aux=True, full_ref="numpy", jit=False, lanes=1):
    """fcn / deeplabv3 on the dilated ResNet (reference tests/test_models/test_fcn.py:16-28, test_deeplabv3.py): (aux, out) at the
    input resolution vs the oracle on the same synthetic checkpoint."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd.models.classification import resnet as R
        sd = S.segmentation_state(1, kind, layers, classes, aux)
        x = S.synthetic_images(B, size, seed=0)
        bb = lambda: R._resnet(R._ResNetBottleneck, list(layers), None, replace_stride_with_dilation=[False, True, True])
        build = eqv.models.fcn if kind == "fcn" else eqv.models.deeplabv3
        tap = (lambda m: [m.layer3, m.layer4]) if aux else (lambda m: [m.layer4])
        fac = lambda torch_weights=None, **kw: build(num_classes=classes, backbone=bb(), intermediate_layers=tap,
                                                     aux_in_channels=1024 if aux else None, torch_weights=torch_weights)
        net = _load(fac, sd)
        if jit:
            with eqv.precision(dtype):
                fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), lanes=lanes)
                for _ in range(4):
                    got = fwd(net, x, _keys(B))          # trace, capture, replays
            torch.cuda.synchronize()
            entry = fwd._entries()[0]
            lane_note = len(entry.lane_calls) if entry.lane_calls else 1
        else:
            got = _run(net, x, dtype)
            lane_note = 1
        g_aux, g_out = got
        if full_ref == "torch":
            r_aux, r_out = TR.segmentation_forward(sd, x, kind, layers, aux)
            r_out = r_out.numpy()
            r_aux = r_aux.numpy() if aux else None
        else:
            pairs = [OM.segmentation_forward(sd, im, kind, layers, aux) for im in x]
            r_out = np.stack([p[1] for p in pairs])
            r_aux = np.stack([p[0] for p in pairs]) if aux else None
        tol = 1e-2 if dtype == "bf16" else 1e-3
        info = _cmp(g_out.cpu().numpy(), r_out, tol)
        info["shape_ok"] = tuple(g_out.shape) == (B, classes, size, size)
        if aux:
            ia = _cmp(g_aux.cpu().numpy(), r_aux, tol)
            info["aux_err"] = ia.get("err")
            info["ok"] = info["ok"] and ia["ok"]
        else:
            info["ok"] = info["ok"] and g_aux is None
        info["ok"] = info["ok"] and info["shape_ok"] and lane_note == lanes
        info["lanes"] = lane_note
        return info
    return run


def vit_case(img, patch, dim, depth, heads, B, classes=10, dtype="bf16", attn=False, full_ref="numpy"):
    def run():
        import eqxvision_amd as eqv
        sd = S.vit_state(1, img, patch, dim, depth, heads, 4, classes)
        x = S.synthetic_images(B, img, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.models.VisionTransformer(**kw) if torch_weights is None else \
            eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
        net = _load(fac, sd, img_size=img, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads,
                    num_classes=classes)
        if attn:
            with eqv.precision(dtype):
                got = eqv.vmap(net.get_last_self_attention)(x, key=_keys(B)).cpu().numpy()
            ref = TR.vit_last_self_attention(sd, x, patch, heads, depth).numpy()
            return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.vit_forward(sd, x, patch, heads, depth).numpy()
            extra = {}
        else:
            ref = O.vmap(lambda im: OM.vit_forward(sd, im, patch, heads, depth))(x)
            extra = {}
            if dtype == "bf16":
                emu = O.vmap(lambda im: OM.vit_forward(sd, im, patch, heads, depth, bf16=True))(x)
                extra["err_vs_bf16_emulation"] = float(np.abs(got - emu).max())
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3, extra)
    return run


def swin_case(size, embed, depths, heads, B, classes=10, dtype="bf16", full_ref="numpy"):
    def run():
        import warnings
        import eqxvision_amd as eqv
        sd = S.swin_state(1, (4, 4), embed, depths, heads, (7, 7), 4.0, classes)
        x = S.synthetic_images(B, size, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.SwinTransformer(**kw), torch_weights)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = _load(fac, sd, patch_size=[4, 4], embed_dim=embed, depths=list(depths), num_heads=list(heads),
                        window_size=[7, 7], num_classes=classes)
        got = _run(net, x, dtype).cpu().numpy()
        if full_ref == "torch":
            ref = TR.swin_forward(sd, x, (4, 4), depths, heads, (7, 7)).numpy()
        else:
            ref = O.vmap(lambda im: OM.swin_forward(sd, im, (4, 4), depths, heads, (7, 7)))(x)
        return _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3)
    return run


def swin_train_case(size=56, embed=32, depths=(2, 2), heads=(2, 4), B=4, classes=10, sd_prob=0.5):
    """Swin outside inference mode: stochastic depth is live (DropPath(mode="local"), one draw per channel, rate growing with the
    block index: swin.py:545, 572-578, 726-730); keys split as the reference splits them -> the oracle drops the same channels."""
    def run():
        import warnings
        import eqxvision_amd as eqv
        sd = S.swin_state(1, (4, 4), embed, depths, heads, (7, 7), 4.0, classes)
        x = S.synthetic_images(B, size, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.SwinTransformer(**kw), torch_weights)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = _load(fac, sd, patch_size=[4, 4], embed_dim=embed, depths=list(depths), num_heads=list(heads),
                        window_size=[7, 7], num_classes=classes, stochastic_depth_prob=sd_prob)
        net = eqv.tree_inference(net, False)
        keys = eqv.random.split(eqv.random.PRNGKey(31), B)
        with eqv.precision("bf16"):
            got = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))(net, x, keys).cpu().numpy()
        ref = np.stack([OM.swin_forward(sd, x[i], (4, 4), depths, heads, (7, 7), bf16=True, key=keys[i],
                                        stochastic_depth_prob=sd_prob) for i in range(B)])
        inf = np.stack([OM.swin_forward(sd, x[i], (4, 4), depths, heads, (7, 7), bf16=True) for i in range(B)])
        out = _cmp(got, ref, 1e-2)
        out["differs_from_inference"] = float(np.abs(got - inf).max())
        out["ok"] = out["ok"] and out["differs_from_inference"] > 5 * max(out["err"], 1e-3)
        return out
    return run


def swin_dropout_case(training=True, size=56, embed=64, depths=(2, 2), heads=(2, 4), B=3, classes=10, sd_prob=0.2, dropout=0.1,
                      attention_dropout=0.2, effect=5.0):
    """Swin with dropout / attention_dropout > 0: the reference's `_func_dropout` draws on the window-layout probabilities and
    projection output in EVERY mode (swin.py:17-20, 227, 233), the MLP's Dropouts and stochastic depth in training mode only; masks
    from the same keys as the oracle's.  32 channels per head: the MFMA window-attention kernel's dropout variant."""
    def run():
        import warnings
        import eqxvision_amd as eqv
        sd = S.swin_state(1, (4, 4), embed, depths, heads, (7, 7), 4.0, classes)
        x = S.synthetic_images(B, size, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.SwinTransformer(**kw), torch_weights)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = _load(fac, sd, patch_size=[4, 4], embed_dim=embed, depths=list(depths), num_heads=list(heads),
                        window_size=[7, 7], num_classes=classes, stochastic_depth_prob=sd_prob, dropout=dropout,
                        attention_dropout=attention_dropout)
        net = eqv.tree_inference(net, not training)
        keys = eqv.random.split(eqv.random.PRNGKey(37), B)
        with eqv.precision("bf16"):
            f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))
            got = f(net, x, keys).cpu().numpy()
            again = f(net, x, eqv.random.split(eqv.random.PRNGKey(38), B)).cpu().numpy()     # fresh keys -> fresh masks (never replayed)
            try:
                eqv.vmap(net, axis_name="batch")(x)
                keyless = "no error"
            except RuntimeError:
                keyless = "RuntimeError"
        ref = np.stack([OM.swin_forward(sd, x[i], (4, 4), depths, heads, (7, 7), bf16=True, key=keys[i], stochastic_depth_prob=sd_prob,
                                        dropout=dropout, attention_dropout=attention_dropout, training=training) for i in range(B)])
        inf = np.stack([OM.swin_forward(sd, x[i], (4, 4), depths, heads, (7, 7), bf16=True) for i in range(B)])
        out = _cmp(got, ref, 1e-2)
        out["differs_from_no_dropout"] = float(np.abs(got - inf).max())
        out["differs_with_other_keys"] = float(np.abs(got - again).max())
        out["keyless_call"] = keyless
        out["ok"] = (out["ok"] and out["differs_from_no_dropout"] > effect * max(out["err"], 1e-3)
                     and out["differs_with_other_keys"] > effect * max(out["err"], 1e-3) and keyless == "RuntimeError")
        return out
    return run


def vit_train_case(img=32, patch=8, dim=64, depth=4, heads=2, B=4, classes=10, rate=0.6):
    """ViT outside inference mode with drop_path_rate > 0: DropPath(mode="global") per block, rates linspace(0, rate, depth), keys
    split per block and 4 ways inside (vit.py:148-156, 236-246, 267-271) -> the oracle drops the same residual branches."""
    def run():
        import eqxvision_amd as eqv
        sd = S.vit_state(1, img, patch, dim, depth, heads, 4, classes)
        x = S.synthetic_images(B, img, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
        net = eqv.tree_inference(_load(fac, sd, img_size=img, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads,
                                       num_classes=classes, drop_path_rate=rate), False)
        keys = eqv.random.split(eqv.random.PRNGKey(41), B)
        with eqv.precision("bf16"):
            got = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))(net, x, keys).cpu().numpy()
        ref = np.stack([OM.vit_forward(sd, x[i], patch, heads, depth, bf16=True, key=keys[i], drop_path_rate=rate) for i in range(B)])
        inf = np.stack([OM.vit_forward(sd, x[i], patch, heads, depth, bf16=True) for i in range(B)])
        out = _cmp(got, ref, 1e-2)
        out["differs_from_inference"] = float(np.abs(got - inf).max())
        out["ok"] = out["ok"] and out["differs_from_inference"] > 5 * max(out["err"], 1e-3)
        return out
    return run


def vit_dropout_case(img=32, patch=8, dim=64, depth=3, heads=2, B=4, classes=10, drop=0.1, attn_drop=0.2, path=0.2):
    """ViT outside inference mode with drop_rate / attn_drop_rate / drop_path_rate > 0: the attention probabilities' dropout inside
    the attention kernel, the projection's per-sample dropout and the MLP's per-TOKEN dropouts (the reference vmaps the MLP over the
    tokens with split keys, vit.py:155) -- the oracle draws the same masks from the same keys."""
    def run():
        import eqxvision_amd as eqv
        sd = S.vit_state(1, img, patch, dim, depth, heads, 4, classes)
        x = S.synthetic_images(B, img, seed=0)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
        net = eqv.tree_inference(_load(fac, sd, img_size=img, patch_size=patch, embed_dim=dim, depth=depth, num_heads=heads,
                                       num_classes=classes, drop_rate=drop, attn_drop_rate=attn_drop, drop_path_rate=path), False)
        keys = eqv.random.split(eqv.random.PRNGKey(43), B)
        with eqv.precision("bf16"):
            got = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))(net, x, keys).cpu().numpy()
            try:
                eqv.vmap(net, axis_name="batch")(x)
                keyless = "no error"
            except RuntimeError as e:
                keyless = "RuntimeError" if "requires a key" in str(e) else repr(e)
        ref = np.stack([OM.vit_forward(sd, x[i], patch, heads, depth, bf16=True, key=keys[i], drop_path_rate=path, drop_rate=drop,
                                       attn_drop_rate=attn_drop) for i in range(B)])
        inf = np.stack([OM.vit_forward(sd, x[i], patch, heads, depth, bf16=True) for i in range(B)])
        out = _cmp(got, ref, 1e-2)
        out["differs_from_inference"] = float(np.abs(got - inf).max())
        out["keyless_call"] = keyless
        out["ok"] = out["ok"] and out["differs_from_inference"] > 5 * max(out["err"], 1e-3) and keyless == "RuntimeError"
        return out
    return run


def cna_family_train_case():
    """Training-mode BatchNorm inside the ConvNormActivation families (depthwise / odd-width / squeeze-excitation stacks: the
    Sequential peephole may not fold a BatchNorm that is not in inference mode): a reduced MobileNetV2 takes two training steps;
    every BatchNorm's running statistics must have moved (EMA: 0.01 of the batch moments per step), and an INFERENCE forward of the
    same modules must equal the oracle's inference forward on the exported state (statistics refolded into every kernel's
    epilogue: the folds are keyed on the shared state's version)."""
    def run():
        import eqxvision_amd as eqv
        st = ((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 2, 1))
        sd = S.mobilenet_v2_state(1, 10, st, last=64)
        from eqxvision_amd._module import tree_at
        from eqxvision_amd.layers import ConvNormActivation

        def fac(torch_weights=None, **k):
            net = eqv.models.MobileNetV2(**k)
            cin = net.features.layers[-1].layers[0].in_channels
            net = tree_at(lambda m: (m.features.layers[-1], m.classifier.layers[-1]), net,
                          (ConvNormActivation(cin, 64, kernel_size=1, key=eqv.random.PRNGKey(5)),
                           eqv.nn.Linear(64, 10, key=eqv.random.PRNGKey(6))))
            return eqv.utils.load_torch_weights(net, torch_weights)
        net_inf = _load(fac, sd, num_classes=10, inverted_residual_setting=[list(r) for r in st])
        x = S.synthetic_images(6, 64, seed=0)
        with eqv.precision("bf16"):
            before = eqv.vmap(net_inf, axis_name="batch")(x, key=_keys(6)).cpu().numpy()      # folds of the LOADED statistics cached
            net = eqv.tree_inference(net_inf, False)
            for step in range(2):
                out = eqv.vmap(net, axis_name="batch")(S.synthetic_images(6, 64, seed=step + 1), key=_keys(6))
            torch.cuda.synchronize()
            after = eqv.vmap(net_inf, axis_name="batch")(x, key=_keys(6)).cpu().numpy()
        sd2 = {k: np.asarray(v) for k, v in eqv.utils.state_dict(net_inf).items()}
        moved = [float(np.abs(sd2[k] - sd[k]).max()) for k in sd if k.endswith("running_mean")]
        ref = np.stack([OM.mobilenet_v2_forward(sd2, im, st, bf16=True) for im in x])
        info = _cmp(after, ref, 1e-2)
        info.update(batchnorms=len(moved), min_running_mean_shift=min(moved), logits_changed=float(np.abs(after - before).max()),
                    train_logits_finite=bool(torch.isfinite(out).all()))
        info["ok"] = info["ok"] and min(moved) > 0.0 and info["logits_changed"] > 0.0 and info["train_logits_finite"]
        return info
    return run


def bn_first_steps_case(C=40, hw=9, B=5):
    """A fresh BatchNorm (no running statistics: what the reference constructs) through its first three training steps, as a
    module on a map: step 1 takes the reference's literal two passes and running = batch; steps 2-3 the single pass centred on the
    running mean and the EMA -- outputs and statistics against oracle.np_ops.batchnorm_train each step."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import nn
        rng = np.random.Generator(np.random.PCG64(9))
        bn = nn.BatchNorm(C, axis_name="batch")
        object.__setattr__(bn, "weight", rng.uniform(0.5, 1.5, C).astype(np.float32))
        object.__setattr__(bn, "bias", (0.1 * rng.standard_normal(C)).astype(np.float32))
        running, errs, serr = None, [], []
        with eqv.precision("fp32"):
            for step in range(3):
                x = (rng.standard_normal((B, C, hw, hw)) * rng.uniform(0.5, 3.0, (1, C, 1, 1)) + rng.uniform(-4, 4, (1, C, 1, 1))
                     + 0.3 * step).astype(np.float32)
                got = eqv.vmap(bn, axis_name="batch")(x, key=_keys(B)).cpu().numpy()
                want, running = O.batchnorm_train(x, bn.weight, bn.bias, running, step == 0)
                errs.append(float(np.abs(got - want).max()))
                m, v = bn.state_index.value
                serr.append(max(float(np.abs(np.asarray(m) - running[0]).max()), float(np.abs(np.asarray(v) - running[1]).max())))
        ok = max(errs) < 1e-4 and max(serr) < 1e-4 and not bn.first_time_index.value
        return {"ok": ok, "err": max(errs), "lim": 1e-4, "output_errs": errs, "running_stat_errs": serr}
    return run


def jit_case():
    """filter_jit: the Python body runs once; replays (call 2 = hipGraph capture, call 3 = graph launch)
    with NEW inputs must equal eager results (reference semantics: tests/test_models/test_vit.py:35)."""
    def run():
        import eqxvision_amd as eqv
        sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
        blk = eqv.models.classification.resnet._ResNetBottleneck
        fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], torch_weights, **kw)
        net = _load(fac, sd, num_classes=10)
        count = [0]

        def body(n, im, k):
            count[0] += 1
            return eqv.vmap(n, axis_name="batch")(im, key=k)

        fwd = eqv.filter_jit(body)
        errs = []
        for seed in (0, 1, 2, 3):
            x = S.synthetic_images(4, 64, seed=seed)
            got = fwd(net, x, _keys(4)).cpu().numpy()
            ref = eqv.vmap(net, axis_name="batch")(x, key=_keys(4)).cpu().numpy()
            errs.append(float(np.abs(got - ref).max()))
        return {"ok": max(errs) == 0.0 and count[0] == 1, "errs": errs, "traces": count[0]}
    return run


def lanes_case(model="resnet", lanes=2, B=6):
    """filter_jit(lanes=k): sub-batches captured as parallel graph branches must give exactly what the single-lane
    path gives for the same sub-batch shapes, for fresh inputs on every replay; odd batches fall back to one lane."""
    def run():
        import eqxvision_amd as eqv
        if model == "resnet":
            sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
            blk = eqv.models.classification.resnet._ResNetBottleneck
            fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], torch_weights, **kw)
            net = _load(fac, sd, num_classes=10)
            size = 64
        else:
            sd = S.vit_state(1, 64, 8, 32, 2, 2, 10)
            net = _load(lambda torch_weights=None, **kw: eqv.models.VisionTransformer(
                img_size=64, patch_size=8, embed_dim=32, depth=2, num_heads=2, **kw), sd, num_classes=10)
            size = 64
        fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), lanes=lanes)
        errs = []
        step = B // lanes
        for seed in (0, 1, 2, 3):
            x = S.synthetic_images(B, size, seed=seed)
            got = fwd(net, x, _keys(B)).cpu().numpy()
            ref = np.concatenate([eqv.vmap(net, axis_name="batch")(x[l * step:(l + 1) * step], key=_keys(step)).cpu().numpy()
                                  for l in range(lanes)])
            errs.append(float(np.abs(got - ref).max()))
        c = fwd._entries()[0]
        x = S.synthetic_images(B + 1, size, seed=9)          # B+1 not divisible: one lane, still correct
        got = fwd(net, x, _keys(B + 1)).cpu().numpy()
        ref = eqv.vmap(net, axis_name="batch")(x, key=_keys(B + 1)).cpu().numpy()
        errs.append(float(np.abs(got - ref).max()))
        return {"ok": max(errs) == 0.0 and c.lane_calls is not None and len(c.lane_calls) == lanes and c.graph is not None,
                "errs": errs, "lanes": None if c.lane_calls is None else len(c.lane_calls)}
    return run


def fresh_inputs_case():
    """A caller that hands filter_jit a NEW resident device tensor every step (the usual data-loader pattern): the first
    MAX_INPLACE_VARIANTS addresses get zero-copy graphs, every later one reuses ONE variant that owns its input buffer
    (device-to-device copy per call) -- no retrace, no unbounded pinning -- and every result equals the eager one."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import transforms as T
        sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
        blk = eqv.models.classification.resnet._ResNetBottleneck
        fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], torch_weights, **kw)
        net = _load(fac, sd, num_classes=10)
        count = [0]

        def body(n, im, k):
            count[0] += 1
            return eqv.vmap(n, axis_name="batch")(im, key=k)

        fwd = eqv.filter_jit(body, lanes=2)
        errs, held = [], []
        for seed in range(9):
            x = torch.as_tensor(S.synthetic_images(4, 64, seed=seed)).cuda()
            held.append(x)                                   # keep them all alive: nine distinct addresses
            got = fwd(net, x, _keys(4)).cpu().numpy()
            ref = eqv.vmap(net, axis_name="batch")(x, key=_keys(4)).cpu().numpy()
            errs.append(float(np.abs(got - ref).max()))
        for seed in (0, 1):                                  # the first two buffers again: their zero-copy graphs replay
            x = held[seed]
            x.copy_(torch.as_tensor(S.synthetic_images(4, 64, seed=20 + seed)))
            got = fwd(net, x, _keys(4)).cpu().numpy()
            ref = eqv.vmap(net, axis_name="batch")(x, key=_keys(4)).cpu().numpy()
            errs.append(float(np.abs(got - ref).max()))
        ent = fwd._entries()
        owned = [c for c in ent if c.own_resident]
        return {"ok": max(errs) == 0.0 and len(ent) == T.MAX_INPLACE_VARIANTS + 1 and len(owned) == 1 and
                count[0] == 2 * (T.MAX_INPLACE_VARIANTS + 1) and owned[0].graph is not None,
                "errs": errs, "entries": len(ent), "traces": count[0], "owned_replays": owned[0].replays if owned else -1}
    return run


def full_batch_case(model, B, lanes=2, guard=None):
    """BASELINE.json's full configuration (resnet50 / vit_base B=256, swin_t B=128) on the bench path -- filter_jit, hipGraph
    replay, graph lanes -- checked through a size-independent property: the batch is 8 distinct images tiled B/8 times, so
    (a) rows 0..7 must match the fp32 torch restatement of the same 8 images within the bf16 tolerance and (b) every
    row r must agree with row r % 8 (same image, different position in the batch / tile / lane)."""
    def run():
        import warnings
        import eqxvision_amd as eqv
        x8 = S.synthetic_images(8, 224, seed=3)
        if model == "resnet50":
            sd = S.resnet_state(1, "bottleneck", (3, 4, 6, 3), 1000)
            net = _load(eqv.models.resnet50, sd, num_classes=1000)
            ref = TR.resnet_forward(sd, x8, "bottleneck", (3, 4, 6, 3)).numpy()
        elif model == "vit_base":
            sd = S.vit_state(1, 224, 16, 768, 12, 12, 4, 1000)
            fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
            net = _load(fac, sd, img_size=224, patch_size=16, embed_dim=768, depth=12, num_heads=12, num_classes=1000)
            ref = TR.vit_forward(sd, x8, 16, 12, 12).numpy()
        else:
            sd = S.swin_state(1, (4, 4), 96, (2, 2, 6, 2), (3, 6, 12, 24), (7, 7), 4.0, 1000)
            fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.SwinTransformer(**kw), torch_weights)
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                net = _load(fac, sd, patch_size=[4, 4], embed_dim=96, depths=[2, 2, 6, 2], num_heads=[3, 6, 12, 24],
                            window_size=[7, 7], num_classes=1000)
            ref = TR.swin_forward(sd, x8, (4, 4), (2, 2, 6, 2), (3, 6, 12, 24), (7, 7)).numpy()
        x = torch.as_tensor(np.asarray(x8)).repeat(B // 8, 1, 1, 1).cuda()
        with eqv.precision("bf16"):
            fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), lanes=lanes)
            for _ in range(3):                      # trace, capture, replay
                got = fwd(net, x, _keys(B))
        got = got.float().cpu().numpy()
        out = _cmp(got[:8], ref, 1e-2)
        rep = np.abs(got.reshape(B // 8, 8, -1) - got[None, :8]).max()
        out["replica_max_diff"] = float(rep)
        out["replicas_bit_identical"] = bool(rep == 0.0)
        out["ok"] = bool(out["ok"] and rep <= out["lim"])
        if guard is not None:          # a margin under the 1e-2 bar: a new bf16 fusion must not eat it silently (swin_t sits at 8.3e-3)
            out["guard"] = guard
            out["ok"] = bool(out["ok"] and out["err"] <= guard)
        return out
    return run


def conv_norm_act_case(cin, cout, hw, B, kernel=3, stride=1, dilation=1, norm="bn", act="relu", dtype="bf16", seed=0):
    """`eqxvision.layers.ConvNormActivation` as a MODULE on the device (reference layers/conv_norm_activation.py:18-86): the
    Sequential [Conv2d, BatchNorm, Lambda(relu)] runs as ONE fused launch (nn.Sequential peephole) and must match the oracle's
    conv -> BatchNorm(inference) -> activation.  (cin, cout, hw) = (3, 4, 5) is the reference's own test
    (tests/test_layers.py:83-92: output shape (4, 5, 5), every value >= 0)."""
    def run():
        import functools
        import eqxvision_amd as eqv
        from eqxvision_amd import nn
        rng = np.random.Generator(np.random.PCG64(seed))
        norm_layer = {"bn": nn.BatchNorm, "bn_partial": functools.partial(nn.BatchNorm, eps=1e-3), "none": None}[norm]
        act_layer = {"relu": nn.relu, "gelu": nn.gelu, "none": None}[act]
        m = eqv.layers.ConvNormActivation(cin, cout, kernel_size=kernel, stride=stride, dilation=dilation, norm_layer=norm_layer,
                                          activation_layer=act_layer, key=eqv.random.PRNGKey(seed + 1))
        conv = m.layers[0]
        bn = m.layers[1] if norm != "none" else None
        w = np.asarray(conv.weight, np.float32)
        bias = None if conv.bias is None else np.asarray(conv.bias, np.float32).reshape(-1)
        if bn is not None:                       # non-trivial running statistics + affine, set like load_torch_weights does
            g = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            g[::3] *= -1.0
            b = (0.1 * rng.standard_normal(cout)).astype(np.float32)
            mean = (0.1 * rng.standard_normal(cout)).astype(np.float32)
            var = rng.uniform(0.5, 1.5, cout).astype(np.float32)
            object.__setattr__(bn, "weight", g)
            object.__setattr__(bn, "bias", b)
            bn.state_index.value = (mean, var)
        m = eqv.tree_inference(m, True)
        x = rng.standard_normal((B, cin, hw, hw)).astype(np.float32)
        with eqv.precision(dtype):
            got = eqv.vmap(m, axis_name="batch")(x, key=_keys(B))
        torch.cuda.synchronize()
        from eqxvision_amd import _lib
        kern = _lib.last_kernel()
        got = got.cpu().numpy()
        q = O.bf16_round if dtype == "bf16" else (lambda a_: np.asarray(a_, np.float32))
        pad = (kernel - 1) // 2 * dilation
        ref = np.stack([O.conv2d(q(x[i]), q(w), bias, stride, pad, dilation) for i in range(B)])
        if bn is not None:
            eps = bn.eps
            ref = np.stack([O.batchnorm_inference(ref[i], g, b, mean, var, eps) for i in range(B)])
        if act == "relu":
            ref = O.relu(ref)
        elif act == "gelu":
            ref = O.gelu_tanh(ref)
        info = _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3, scaled=True)
        info["kernel"] = kern
        Ho = (hw + 2 * pad - dilation * (kernel - 1) - 1) // stride + 1
        info["shape_ok"] = tuple(got.shape) == (B, cout, Ho, Ho) and m.out_channels == cout
        info["nonneg"] = bool((got >= 0).all()) if act == "relu" else None
        info["ok"] = bool(info["ok"] and info["shape_ok"] and (info["nonneg"] is not False))
        return info
    return run


def layer_getter_case(seed=0, dtype="bf16"):
    """`intermediate_layer_getter` (reference experimental.py:35-88) on the device: (a) an nn.Sequential addressed by index --
    each intermediate vs the oracle's conv chain; (b) a ResNet addressed through attributes (tree_at) -- final logits equal the
    plain model's, intermediates have the reference's (C,H,W) shapes; wrapping must not change the result."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import nn
        from eqxvision_amd.experimental import intermediate_layer_getter
        rng = np.random.Generator(np.random.PCG64(seed))
        k = eqv.random.split(eqv.random.PRNGKey(seed + 3), 3)
        seq = nn.Sequential([nn.Conv2d(3, 64, 3, padding=1, key=k[0]), nn.Lambda(nn.relu), nn.Conv2d(64, 64, 3, padding=1, key=k[1]),
                             nn.Lambda(nn.relu), nn.Conv2d(64, 32, 1, key=k[2])])
        g = intermediate_layer_getter(seq, lambda m: [1, 3])
        B, hw = 3, 20
        x = rng.standard_normal((B, 3, hw, hw)).astype(np.float32)
        q = O.bf16_round if dtype == "bf16" else (lambda a_: np.asarray(a_, np.float32))
        with eqv.precision(dtype):
            out, inter = eqv.vmap(g, axis_name="batch")(x, key=_keys(B))
        torch.cuda.synchronize()

        def conv(v, c, pad):
            return np.stack([O.conv2d(v[i], q(np.asarray(c.weight)), np.asarray(c.bias).reshape(-1), 1, pad) for i in range(B)])
        r1 = q(O.relu(conv(q(x), seq.layers[0], 1)))
        r2 = q(O.relu(conv(r1, seq.layers[2], 1)))
        r3 = conv(r2, seq.layers[4], 0)
        tol = 1e-2 if dtype == "bf16" else 1e-3
        i1 = _cmp(inter[0].cpu().numpy(), r1, tol, scaled=True)
        i2 = _cmp(inter[1].cpu().numpy(), r2, 2 * tol, scaled=True)
        i3 = _cmp(out.cpu().numpy(), r3, 2 * tol, scaled=True)
        # (b) attribute targets on a ResNet
        sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
        blk = eqv.models.classification.resnet._ResNetBottleneck
        fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], torch_weights, **kw)
        net = _load(fac, sd, num_classes=10)
        xi = S.synthetic_images(2, 64, seed=0)
        plain = _run(net, xi, dtype).cpu().numpy()
        g2 = intermediate_layer_getter(net, lambda m: [m.layer1, m.layer3])
        with eqv.precision(dtype):
            o2, in2 = eqv.vmap(g2, axis_name="batch")(xi, key=_keys(2))
        torch.cuda.synchronize()
        shapes_ok = tuple(in2[0].shape) == (2, 256, 16, 16) and tuple(in2[1].shape) == (2, 1024, 4, 4)
        same = float(np.abs(o2.cpu().numpy() - plain).max())
        info = {"ok": bool(i1["ok"] and i2["ok"] and i3["ok"] and shapes_ok and same <= 2e-3), "err": max(i1["err"], i2["err"], i3["err"]),
                "seq_errs": [i1["err"], i2["err"], i3["err"]], "resnet_shapes_ok": shapes_ok, "resnet_wrapped_vs_plain": same}
        return info
    return run


def pth_reader_case():
    """A checkpoint ingested by the torch-free reader (eqxvision_amd/pth.py) gives the same model as torch.load would: the whole
    load_torch_weights -> device forward path with `torch.load` made unavailable."""
    def run():
        import eqxvision_amd as eqv
        import eqxvision_amd.utils as U
        sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
        blk = eqv.models.classification.resnet._ResNetBottleneck
        fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], torch_weights, **kw)
        real_load = torch.load
        torch.load = None                                    # any use would raise TypeError
        try:
            net = _load(fac, sd, num_classes=10)
        finally:
            torch.load = real_load
        x = S.synthetic_images(2, 64, seed=0)
        got = _run(net, x, "bf16").cpu().numpy()
        ref = O.vmap(lambda im: OM.resnet_forward(sd, im, "bottleneck", (1, 1, 1, 1)))(x)
        return _cmp(got, ref, 1e-2)
    return run


def _hot_model(name, sd):
    """(factory-loaded inference model, torch-restatement forward) of one of the three full-size hot models on checkpoint `sd`."""
    import warnings
    import eqxvision_amd as eqv
    if name == "resnet50":
        return _load(eqv.models.resnet50, sd), lambda x: TR.resnet_forward(sd, x).numpy()
    if name == "vit_base":
        return _load(eqv.models.vit_base, sd, num_classes=1000), lambda x: TR.vit_forward(sd, x).numpy()
    if name == "alexnet":
        return _load(eqv.models.alexnet, sd), lambda x: TR.alexnet_forward(sd, x).numpy()
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        return _load(eqv.models.swin_t, sd), lambda x: TR.swin_forward(sd, x).numpy()


_HOT_STATE = {"resnet50": lambda: S.resnet_state(1), "vit_base": lambda: S.vit_state(1), "swin_t": lambda: S.swin_state(1),
              "alexnet": lambda: S.alexnet_state(1, 1000)}
_HEAD = {"resnet50": "fc", "vit_base": "fc", "swin_t": "head"}


def committed_golden_case(name, dtype="bf16"):
    """The HIP path against the COMMITTED vectors of tests/golden/hotpath_small.npz (generated by tests/golden/make_golden.py from
    the torch restatement) -- not against an oracle regenerated in this process: first 16 logits and the L2 norm
    of the logits of the full-size model at B = 2, 224 px."""
    def run():
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_small.npz"))
        net, _ = _hot_model(name, _HOT_STATE[name]())
        x = S.synthetic_images(2, 224, seed=0)
        got = _run(net, x, dtype).cpu().numpy()
        head, l2 = gold[f"{name}_224_logits_head16"], gold[f"{name}_224_logits_l2"]
        tol = 1e-2 if dtype == "bf16" else 1e-3
        info = _cmp(got[:, :16], head, tol)
        l2err = float(np.abs(np.linalg.norm(got.astype(np.float64), axis=1) - l2).max())
        info["l2_err"] = l2err
        info["ok"] = bool(info["ok"] and l2err <= tol * np.sqrt(got.shape[1]))       # |  ||a|| - ||b||  | <= ||a - b|| <= tol * sqrt(classes)
        return info
    return run


def committed_small_golden_case(dtype="bf16"):
    """The reduced models of tests/golden/hotpath_small.npz (the same bottleneck / attention / window code at small sizes), HIP
    path vs the committed logits."""
    def run():
        import warnings
        import eqxvision_amd as eqv
        gold = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hotpath_small.npz"))
        tol = 1e-2 if dtype == "bf16" else 1e-3
        R = eqv.models.classification.resnet
        errs = {}
        x64 = S.synthetic_images(2, 64, seed=0)
        net = _load(lambda torch_weights=None, **kw: R._resnet(R._ResNetBottleneck, [1, 1, 1, 1], torch_weights, **kw),
                    S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10), num_classes=10)
        errs["resnet_bottleneck_1111_64px"] = float(np.abs(_run(net, x64, dtype).cpu().numpy() - gold["resnet_bottleneck_1111_64px"]).max())
        net = _load(eqv.models.resnet18, S.resnet_state(1, "basic", (2, 2, 2, 2), 10), num_classes=10)
        errs["resnet18_64px"] = float(np.abs(_run(net, x64, dtype).cpu().numpy() - gold["resnet18_64px"]).max())
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
        net = _load(fac, S.vit_state(1, 32, 8, 64, 2, 2, 4, 10), img_size=32, patch_size=8, embed_dim=64, depth=2, num_heads=2, num_classes=10)
        errs["vit_32px_p8_d64_h2_depth2"] = float(np.abs(_run(net, S.synthetic_images(3, 32, seed=0), dtype).cpu().numpy()
                                                         - gold["vit_32px_p8_d64_h2_depth2"]).max())
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.SwinTransformer(**kw), torch_weights)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = _load(fac, S.swin_state(1, (4, 4), 32, (2, 2), (2, 4), (7, 7), 4.0, 10), patch_size=[4, 4], embed_dim=32, depths=[2, 2],
                        num_heads=[2, 4], window_size=[7, 7], num_classes=10)
        errs["swin_56px_e32_d22"] = float(np.abs(_run(net, S.synthetic_images(2, 56, seed=0), dtype).cpu().numpy() - gold["swin_56px_e32_d22"]).max())
        return {"ok": max(errs.values()) <= tol, "err": max(errs.values()), "lim": tol, "errs": errs}
    return run


def large_logit_case(name, target=15.0, B=2, dtype="bf16"):
    """Logits at the scale of a TRAINED classifier (round-2 review, weak 2): the synthetic checkpoints give max|logit| of 0.6 - 2,
    pretrained ones O(10).  The classifier head (weight and bias) of the synthetic checkpoint is scaled so that max|logit| of the
    fp32 restatement is `target`; reports the absolute error and the error relative to max|logit|.  Passes on the RELATIVE
    reading, 1e-2 * max(1, max|logit|) -- bf16 storage keeps 8 mantissa bits, so an absolute 1e-2 cannot hold at |logit| ~ 15 --
    plus identical arg-max; the absolute figure is reported (DESIGN section 4 states which reading the build claims)."""
    def run():
        sd = _HOT_STATE[name]()
        x = S.synthetic_images(B, 224, seed=0)
        _, ref_fn = _hot_model(name, sd)
        s = target / float(np.abs(ref_fn(x)).max())
        h = _HEAD[name]
        sd = type(sd)((k, (v * np.float32(s)).astype(np.float32) if k in (h + ".weight", h + ".bias") else v) for k, v in sd.items())
        net, ref_fn = _hot_model(name, sd)
        ref = ref_fn(x)
        got = _run(net, x, dtype).cpu().numpy()
        info = _cmp(got, ref, 1e-2 if dtype == "bf16" else 1e-3, scaled=True)
        info["abs_err"] = info["err"]
        info["head_scale"] = s
        info["ok"] = bool(info["ok"] and info["argmax_match"])
        return info
    return run


def exported_factory_case(name, B=2, size=224, dtype="bf16"):
    """An exported factory that no other case constructs (round-4 review, weak 7): the published architecture through
    `eqv.models.<name>(torch_weights=...)` at 224 px against the torch restatement on the same synthetic checkpoint.  swin_b's widths
    (128 / 256 / 512 / 1024) and resnet34's basic blocks at full width take dispatch paths of their own."""
    def run():
        import warnings
        import eqxvision_amd as eqv
        kw = {}
        if name in ("swin_s", "swin_b"):
            embed, heads = (96, (3, 6, 12, 24)) if name == "swin_s" else (128, (4, 8, 16, 32))
            sd = S.swin_state(1, (4, 4), embed, (2, 2, 18, 2), heads)
            ref_fn = lambda x: TR.swin_forward(sd, x, (4, 4), (2, 2, 18, 2), heads, (7, 7)).numpy()
        elif name in ("vit_small", "vit_tiny"):
            dim, heads = (384, 6) if name == "vit_small" else (192, 3)
            sd = S.vit_state(1, size, 16, dim, 12, heads)
            ref_fn = lambda x: TR.vit_forward(sd, x, 16, heads, 12).numpy()
            kw = {"num_classes": 1000}
        else:
            block, layers, wpg = {"resnet34": ("basic", (3, 4, 6, 3), 64), "resnet101": ("bottleneck", (3, 4, 23, 3), 64),
                                  "resnet152": ("bottleneck", (3, 8, 36, 3), 64), "wide_resnet50_2": ("bottleneck", (3, 4, 6, 3), 128)}[name]
            sd = S.resnet_state(1, block, layers, 1000, width_per_group=wpg)
            ref_fn = lambda x: TR.resnet_forward(sd, x, block, layers).numpy()
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            net = _load(getattr(eqv.models, name), sd, **kw)
        x = S.synthetic_images(B, size, seed=0)
        got = _run(net, x, dtype).cpu().numpy()
        # every factory is held to the ABSOLUTE bound.  swin_b (widths 128 .. 1024: the un-fused Swin path) sat at 1.15e-2 in round 5
        # and was passed on a relative bound; its block Linears now carry split-precision weights (ops.swin_precise,
        # tests/attrib_swin_bf16.py: weight rounding alone was 1.16e-2); the relative bound is gone from this case.
        return _cmp(got, ref_fn(x), 1e-2 if dtype == "bf16" else 1e-3)
    return run


def join_stream_case(B=8, lanes=2):
    """filter_jit(lanes=2, join="stream") (the lanes are not joined per call: results are futures until `ready()`): five calls with
    the input rewritten in place between `ready()`s must give, bit for bit, what the default (joined) forward gives for the same
    images; the two latest results live in different buffers; misuse raises."""
    def run():
        import eqxvision_amd as eqv
        sd = S.resnet_state(1, "bottleneck", (1, 1, 1, 1), 10)
        blk = eqv.models.classification.resnet._ResNetBottleneck
        fac = lambda torch_weights=None, **kw: eqv.models.classification.resnet._resnet(blk, [1, 1, 1, 1], torch_weights, **kw)
        net = _load(fac, sd, num_classes=10)
        body = lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k)
        with eqv.precision("bf16"):
            fj = eqv.filter_jit(body, clone_outputs=False, lanes=lanes)
            fp = eqv.filter_jit(body, clone_outputs=False, lanes=lanes, join="stream")
            x = torch.from_numpy(S.synthetic_images(B, 64, seed=0)).cuda()
            errs, ptrs = [], []
            for seed in range(1, 7):
                x.copy_(torch.from_numpy(S.synthetic_images(B, 64, seed=seed)).cuda())
                got = fp(net, x, _keys(B))
                fp.ready()
                ref = fj(net, x, _keys(B)).clone()
                torch.cuda.synchronize()
                errs.append(float((got - ref).abs().max()))
                ptrs.append(got.data_ptr())
            # back-to-back calls without ready() in between (the pipelined use), same input: the last result must still be right
            for _ in range(5):
                got = fp(net, x, _keys(B))
            fp.block_until_ready()
            errs.append(float((got - ref).abs().max()))
            refused = 0
            for kw in (dict(lanes=1, join="stream", clone_outputs=False), dict(lanes=2, join="stream"), dict(lanes=2, join="bogus")):
                try:
                    eqv.filter_jit(body, **kw)
                except ValueError:
                    refused += 1
            try:
                fp(net, S.synthetic_images(B, 64, seed=0), _keys(B))       # a host array: not read in place
            except ValueError:
                refused += 1
        ok = max(errs) == 0.0 and len(set(ptrs[-2:])) == 2 and refused == 4
        return {"ok": bool(ok), "err": max(errs), "errs": errs, "two_buffer_sets": len(set(ptrs)) == 2, "misuse_refused": refused}
    return run


def resnet50_rc_case(B=3, size=224):
    """ResNet-50 with the round-6 layer-1 plan (first block output never written, second boundary recomputes it, third block output
    written sub-sampled for the next stage's strided downsample branch; models/classification/resnet.py: _stage_rc) against the same
    network with the plan switched off ("no_chain_rc", "no_chain_sub", "no_chain_res": round 5's launches): both within the bound of the oracle, the two within 2e-3 of
    each other (the recompute changes no rounding point; the next conv1 sums its 256 products in another order), and the launch list
    must show that the plan actually ran."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import _lib
        sd = S.resnet_state(1)
        net = _load(eqv.models.resnet50, sd)
        x = S.synthetic_images(B, size, seed=0)
        ref = TR.resnet_forward(sd, x).numpy()
        rec = []
        old = _lib.set_recording(rec)
        try:
            got = _run(net, x, "bf16").cpu().numpy()
        finally:
            _lib.set_recording(old)
        names = [n for _, _, n in rec]
        for f in ("no_chain_rc", "no_chain_sub", "no_chain_res"):
            _lib.set_flag(f, 1)
        try:
            net2 = _load(eqv.models.resnet50, sd)            # fresh module: no cached decisions
            rec2 = []
            old = _lib.set_recording(rec2)
            try:
                plain = _run(net2, x, "bf16").cpu().numpy()
            finally:
                _lib.set_recording(old)
        finally:
            for f in ("no_chain_rc", "no_chain_sub", "no_chain_res"):
                _lib.set_flag(f, 0)
        names2 = [n for _, _, n in rec2]
        info = _cmp(got, ref, 1e-2)
        info2 = _cmp(plain, ref, 1e-2)
        d = float(np.abs(got - plain).max())
        sub_possible = bool(_lib.load().mv_conv1x1_dual_supported(B * (size // 8) ** 2, 128, 256, 512, 1))     # small batches: no dual kernel
        new = ("mv_conv1x1_chain_rc0_fwd", "mv_conv1x1_chain_rc_fwd", "mv_conv1x1_chain_res_fwd", "mv_conv1x1_chain_sub_fwd")
        ran = all(n in names for n in new[:3])
        off = not any(n in names2 for n in new)
        info.update({"ok": bool(info["ok"] and info2["ok"] and d <= 2e-3 and ran and off), "err_plan_off": info2["err"], "plan_vs_off": d,
                     "plan_ran": ran, "sub_sampled_output_possible": sub_possible, "plan_off_honoured": off, "launches": len(names), "launches_plan_off": len(names2)})
        return info
    return run


def vit_ln_fold_case(B=64, depth=12, stress=False):
    """ViT-B/16 with the LayerNorms of its blocks folded into the GEMM epilogues on either side (round 6: _VitBlock._ln_fold,
    ops.linear_lnout / linear_lnin) against the same network with the fold switched off ("no_ln_fold": LayerNorm launches): both
    within the bound of the fp32 restatement on the 8 distinct images of the batch, the two within 8e-3 of each other (two different sets of bf16 rounding points), and the
    launch list must show the fold where it applies (every norm2 -> fc1, every norm1 -> qkv but the first) and none with the switch."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import _lib
        sd = S.vit_state(1, 224, 16, 768, depth, 12, 4, 1000)
        if stress:      # a residual stream shaped like a trained one: every row offset by ~0.4 of its standard deviation and three
                        # "massive activation" channels at 20 (the fold rounds y, not y - mean, to bf16; the un-fused path rounds LN(y))
            pe = np.array(sd["pos_embed"], np.float32, copy=True)
            pe += np.float32(0.7)
            pe[..., [5, 301, 640]] += np.float32(20.0)
            sd = type(sd)((k, pe if k == "pos_embed" else v) for k, v in sd.items())
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
        kw = dict(img_size=224, patch_size=16, embed_dim=768, depth=depth, num_heads=12, num_classes=1000)
        x8 = S.synthetic_images(8, 224, seed=3)
        ref = TR.vit_forward(sd, x8, 16, 12, depth).numpy()
        x = np.tile(np.asarray(x8), (B // 8, 1, 1, 1))

        def go(flag):
            _lib.set_flag("no_ln_fold", flag)
            try:
                net = _load(fac, sd, **kw)            # fresh module: no cached decisions
                rec = []
                old = _lib.set_recording(rec)
                try:
                    out = _run(net, x, "bf16").cpu().numpy()
                finally:
                    _lib.set_recording(old)
                return out, [n for _, _, n in rec]
            finally:
                _lib.set_flag("no_ln_fold", 0)
        got, names = go(0)
        plain, names2 = go(1)
        info = _cmp(got[:8], ref, 1e-2)
        info2 = _cmp(plain[:8], ref, 1e-2)
        d = float(np.abs(got - plain).max())
        rep = float(np.abs(got.reshape(B // 8, 8, -1) - got[None, :8]).max())
        n_out, n_in, n_ln = names.count("mv_linear_lnout_fwd"), names.count("mv_linear_lnin_fwd"), names.count("mv_layernorm_fwd")
        ran = n_out == 2 * depth and n_in == 2 * depth - 1
        off = not any(n in names2 for n in ("mv_linear_lnout_fwd", "mv_linear_lnin_fwd")) and names2.count("mv_layernorm_fwd") == n_ln + 2 * depth - 1
        info.update({"ok": bool(info["ok"] and info2["ok"] and d <= 8e-3 and rep <= info["lim"] and ran and off), "err_fold_off": info2["err"],
                     "fold_vs_off": d, "replica_max_diff": rep, "fold_ran": ran, "fold_off_honoured": off, "launches": len(names),
                     "launches_fold_off": len(names2)})
        return info
    return run


def vit_split_stream_handover_case(B=64):
    """A folded _VitBlock hands its rows to the next one as two bf16 planes; a block that has no folded form (here: norm2 without an
    affine) must get fp32 rows back (ops.stream_f32: hi + lo) and run the plain launches -- same logits as with the fold switched off."""
    def run():
        import eqxvision_amd as eqv
        from eqxvision_amd import _lib, nn
        sd = S.vit_state(1, 224, 16, 768, 2, 12, 4, 10)
        fac = lambda torch_weights=None, **kw: eqv.utils.load_torch_weights(eqv.models.VisionTransformer(**kw), torch_weights)
        kw = dict(img_size=224, patch_size=16, embed_dim=768, depth=2, num_heads=12, num_classes=10)
        x = np.tile(np.asarray(S.synthetic_images(8, 224, seed=5)), (B // 8, 1, 1, 1))

        def go(flag):
            _lib.set_flag("no_ln_fold", flag)
            try:
                net = _load(fac, sd, **kw)
                object.__setattr__(net.blocks[1], "norm2", nn.LayerNorm(768, eps=net.blocks[1].norm2.eps, elementwise_affine=False))
                rec = []
                old = _lib.set_recording(rec)
                try:
                    out = _run(net, x, "bf16").cpu().numpy()
                finally:
                    _lib.set_recording(old)
                return out, [n for _, _, n in rec]
            finally:
                _lib.set_flag("no_ln_fold", 0)
        got, names = go(0)
        plain, names2 = go(1)
        d = float(np.abs(got - plain).max())
        handed = names.count("mv_linear_lnout_fwd") == 2 and names.count("mv_linear_lnin_fwd") == 1 and "mv_add_fwd" in names
        off = "mv_linear_lnout_fwd" not in names2 and "mv_add_fwd" not in names2
        return {"ok": bool(np.isfinite(got).all() and d <= 8e-3 * max(1.0, float(np.abs(plain).max())) and handed and off), "err": d,
                "refmax": float(np.abs(plain).max()), "split_stream_handed_back_as_fp32": handed, "fold_off_honoured": off}
    return run


def all_cases(full=True):
    c = [("model/resnet_tiny_bottleneck", resnet_case("bottleneck", (1, 1, 1, 1), 64, 2)),
         ("model/resnet18_64px", resnet_case("basic", (2, 2, 2, 2), 64, 2)),
         ("model/stochastic_layers_jax_bitstream", stochastic_layers_case()),
         ("model/alexnet_train_mode_dropout", alexnet_train_case()),
         ("model/swin_train_mode_stochastic_depth", swin_train_case()),
         ("model/vit_train_mode_stochastic_depth", vit_train_case()),
         ("model/swin_dropouts_training_mode", swin_dropout_case(True)),
         ("model/swin_dropouts_inference_mode", swin_dropout_case(False)),
         ("model/swin_attention_dropout_only", swin_dropout_case(True, dropout=0.0, sd_prob=0.0, attention_dropout=0.5, effect=2.0)),
         ("model/vit_train_mode_dropouts", vit_dropout_case()),
         ("model/vit_train_mode_attn_dropout_only", vit_dropout_case(drop=0.0, attn_drop=0.3, path=0.0, depth=2)),
         ("model/vit_train_mode_mlp_dropout_only", vit_dropout_case(drop=0.25, attn_drop=0.0, path=0.0, depth=2)),
         ("model/mobilenet_v2_train_mode_bn_refold", cna_family_train_case()),
         ("model/batchnorm_fresh_first_three_train_steps", bn_first_steps_case()),
         ("model/resnet18_train_mode_bn", resnet_train_case()),
         ("model/resnet18_train_mode_bn_under_filter_jit", resnet_train_case(jit=True)),
         ("model/resnext_tiny_32x4d", resnet_case("bottleneck", (1, 1, 1, 1), 64, 2, groups=32, width_per_group=4)),
         ("model/resnet_tiny_fp32", resnet_case("bottleneck", (1, 1, 1, 1), 64, 2, dtype="fp32")),
         ("model/vit_tiny", vit_case(32, 8, 64, 2, 2, 3)),
         ("model/vit_tiny_fp32", vit_case(32, 8, 64, 2, 2, 2, dtype="fp32")),
         ("model/vit_tiny_last_attn", vit_case(32, 8, 64, 2, 2, 2, attn=True)),
         ("model/mobilenet_v2_reduced", mobilenet_v2_case(((1, 16, 1, 1), (6, 24, 2, 2), (6, 32, 2, 1)), 64, 3, last=64)),
         ("model/mobilenet_v2_reduced_fp32", mobilenet_v2_case(((1, 16, 1, 1), (6, 24, 2, 2)), 32, 2, last=64, dtype="fp32")),
         ("model/mobilenet_v3_small_5rows", mobilenet_v3_case("small", 64, 3, rows=5)),
         ("model/mobilenet_v3_small_5rows_fp32", mobilenet_v3_case("small", 64, 2, rows=5, dtype="fp32")),
         ("model/efficientnet_reduced_mb_and_fused", efficientnet_case(None, 64, 3, stages=((0, 1, 3, 1, 16, 8, 1), (0, 6, 5, 2, 8, 16, 2),
                                                                                        (1, 4, 3, 2, 16, 24, 2), (1, 1, 3, 1, 24, 24, 1)))),
         ("model/efficientnet_reduced_fp32", efficientnet_case(None, 32, 2, dtype="fp32", stages=((0, 1, 3, 1, 16, 8, 1), (0, 6, 3, 2, 8, 16, 2)))),
         ("model/regnet_reduced_y_gw8_24", regnet_case(None, 64, 3, custom=((48, 120), (1, 2), (8, 24), 0.25))),
         ("model/regnet_reduced_x_fp32", regnet_case(None, 32, 2, dtype="fp32", custom=((32, 64), (1, 1), (16, 16), None))),
         ("model/vgg_small_bn_avgpool2x2", vgg_case((16, "M", 32, 32, "M"), True, 56, 3)),
         ("model/vgg_small_fp32", vgg_case((8, "M", 16, "M"), False, 28, 2, dtype="fp32")),
         ("model/vgg_small_c64_128", vgg_case((64, "M", 128, 128, "M"), False, 56, 2)),
         ("model/fcn_tiny_backbone", segmentation_case("fcn", (1, 1, 1, 1), 64, 2)),
         ("model/fcn_tiny_no_aux_fp32", segmentation_case("fcn", (1, 1, 1, 1), 64, 1, dtype="fp32", aux=False)),
         ("model/deeplabv3_tiny_backbone", segmentation_case("deeplabv3", (1, 1, 1, 1), 64, 2)),
         ("model/deeplabv3_tiny_jit_replay", segmentation_case("deeplabv3", (1, 1, 1, 1), 96, 3, jit=True)),
         ("model/deeplabv3_tiny_jit_lanes2_tuple_outputs", segmentation_case("deeplabv3", (1, 1, 1, 1), 96, 4, jit=True, lanes=2)),
         ("model/fcn_tiny_jit_lanes2_no_aux", segmentation_case("fcn", (1, 1, 1, 1), 64, 6, aux=False, jit=True, lanes=2)),
         ("model/swin_tiny", swin_case(56, 32, (2, 2), (2, 4), 2)),
         ("model/swin_tiny_fp32", swin_case(56, 32, (2, 2), (2, 4), 1, dtype="fp32")),
         ("model/conv_norm_act_reference_3_4_5x5", conv_norm_act_case(3, 4, 5, 1)),
         ("model/conv_norm_act_3_4_5x5_fp32", conv_norm_act_case(3, 4, 5, 2, dtype="fp32")),
         ("model/conv_norm_act_64_128_28_mfma", conv_norm_act_case(64, 128, 28, 6, seed=3)),
         ("model/conv_norm_act_128_256_s2_eps", conv_norm_act_case(128, 256, 28, 40, stride=2, norm="bn_partial", seed=4)),
         ("model/conv_norm_act_dil2_gelu_nonorm", conv_norm_act_case(64, 96, 20, 3, dilation=2, norm="none", act="gelu", seed=5)),
         ("model/intermediate_layer_getter", layer_getter_case()),
         ("model/pth_reader_no_torch_load", pth_reader_case()),
         ("model/filter_jit_replay", jit_case()),
         ("model/filter_jit_lanes2_resnet", lanes_case("resnet", 2, 6)),
         ("model/filter_jit_lanes3_resnet", lanes_case("resnet", 3, 6)),
         ("model/filter_jit_fresh_device_inputs", fresh_inputs_case()),
         ("model/filter_jit_lanes2_join_stream_futures", join_stream_case())]
    if full:
        c += [("model/alexnet_features_B2", alexnet_case(2, features_only=True)),
              ("model/alexnet_B4_bf16", alexnet_case(4)),
              ("model/alexnet_B4_fp32", alexnet_case(4, dtype="fp32")),
              ("model/resnet50_B2", resnet_case("bottleneck", (3, 4, 6, 3), 224, 2, classes=1000, full_ref="torch")),
              ("model/resnet50_B3_chained_tail_head", resnet_case("bottleneck", (3, 4, 6, 3), 224, 3, classes=1000, full_ref="torch")),
              ("model/resnet50_B5_odd_200px", resnet_case("bottleneck", (3, 4, 6, 3), 200, 5, classes=1000, full_ref="torch")),
              ("model/resnet50_B3_layer1_recompute_plan_vs_plan_off", resnet50_rc_case()),
              ("model/resnet50_B16_layer1_recompute_plan_vs_plan_off", resnet50_rc_case(B=16)),
              ("model/vit_base_B64_layernorm_fold_vs_fold_off", vit_ln_fold_case()),
              ("model/vit_B64_depth4_layernorm_fold_row_means_and_massive_channels", vit_ln_fold_case(depth=4, stress=True)),
              ("model/vit_split_stream_handed_to_a_block_without_the_fold", vit_split_stream_handover_case()),
              ("model/resnet_2111_160px_B7_mixed_paths", resnet_case("bottleneck", (2, 1, 1, 1), 160, 7, classes=10, full_ref="torch")),
              ("model/resnet_1211_128px_B16_mixed_paths", resnet_case("bottleneck", (1, 2, 1, 1), 128, 16, classes=10, full_ref="torch")),
              ("model/resnext50_32x4d_B2", resnet_case("bottleneck", (3, 4, 6, 3), 224, 2, classes=1000, full_ref="torch",
                                                       groups=32, width_per_group=4)),
              ("model/vit_base_B2", vit_case(224, 16, 768, 12, 12, 2, classes=1000, full_ref="torch")),
              ("model/mobilenet_v2_B4", mobilenet_v2_case(None, 224, 4, classes=1000, full_ref="torch")),
              ("model/mobilenet_v3_large_B4", mobilenet_v3_case("large", 224, 4, classes=1000, full_ref="torch")),
              ("model/mobilenet_v3_small_B3", mobilenet_v3_case("small", 224, 3, classes=1000, full_ref="torch")),
              ("model/lraspp_mobilenet_v3_large_B2", lraspp_case(224, 2)),
              ("model/lraspp_jit_replay_160px_numpy", lraspp_case(160, 2, classes=7, full_ref="numpy", jit=True)),
              ("model/efficientnet_b0_B4", efficientnet_case("b0", 224, 4, classes=1000, full_ref="torch")),
              ("model/efficientnet_v2_s_B2", efficientnet_case("v2_s", 224, 2, classes=1000, full_ref="torch")),
              ("model/regnet_y_400mf_B4", regnet_case("regnet_y_400mf", 224, 4, classes=1000, full_ref="torch")),
              ("model/regnet_x_3_2gf_B2_gw48", regnet_case("regnet_x_3_2gf", 224, 2, classes=1000, full_ref="torch")),
              ("model/vgg11_B2", vgg_case("A", False, 224, 2, classes=1000, full_ref="torch")),
              ("model/vgg16_bn_B1", vgg_case("D", True, 224, 1, classes=1000, full_ref="torch")),
              ("model/fcn_resnet50_B2", segmentation_case("fcn", (3, 4, 6, 3), 224, 2, classes=21, full_ref="torch")),
              ("model/deeplabv3_resnet50_B2", segmentation_case("deeplabv3", (3, 4, 6, 3), 224, 2, classes=21, full_ref="torch")),
              ("model/swin_t_B1", swin_case(224, 96, (2, 2, 6, 2), (3, 6, 12, 24), 1, classes=1000, full_ref="torch")),
              ("model/factory_swin_s_B2", exported_factory_case("swin_s")),
              ("model/factory_swin_b_B2", exported_factory_case("swin_b")),
              ("model/factory_vit_small_B2", exported_factory_case("vit_small")),
              ("model/factory_resnet34_B2", exported_factory_case("resnet34")),
              ("model/factory_resnet101_B2", exported_factory_case("resnet101")),
              ("model/factory_wide_resnet50_2_B2", exported_factory_case("wide_resnet50_2")),
              ("golden/committed_alexnet_224", committed_golden_case("alexnet")),
              ("golden/committed_resnet50_224", committed_golden_case("resnet50")),
              ("golden/committed_vit_base_224", committed_golden_case("vit_base")),
              ("golden/committed_swin_t_224", committed_golden_case("swin_t")),
              ("golden/committed_resnet50_224_fp32", committed_golden_case("resnet50", "fp32")),
              ("golden/committed_vit_base_224_fp32", committed_golden_case("vit_base", "fp32")),
              ("golden/committed_swin_t_224_fp32", committed_golden_case("swin_t", "fp32")),
              ("model/resnet50_B2_fp32", resnet_case("bottleneck", (3, 4, 6, 3), 224, 2, classes=1000, full_ref="torch", dtype="fp32")),
              ("model/vit_base_B2_fp32", vit_case(224, 16, 768, 12, 12, 2, classes=1000, full_ref="torch", dtype="fp32")),
              ("golden/committed_small_models", committed_small_golden_case()),
              ("golden/committed_small_models_fp32", committed_small_golden_case("fp32")),
              ("model/resnet50_logits_at_trained_scale", large_logit_case("resnet50")),
              ("model/vit_base_logits_at_trained_scale", large_logit_case("vit_base")),
              ("model/swin_t_logits_at_trained_scale", large_logit_case("swin_t")),
              ("model/resnet50_B256_full_config", full_batch_case("resnet50", 256)),
              ("model/vit_base_B256_full_config", full_batch_case("vit_base", 256)),
              ("model/swin_t_B128_full_config", full_batch_case("swin_t", 128, guard=9e-3))]
    return c
