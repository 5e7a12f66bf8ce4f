#!/usr/bin/env python
"""Headline benchmark: images/s of the vmap'd inference forward (BASELINE.json configs[1]:
resnet50 bf16 forward, batch 256 per MI355X; `--model vit_base` = configs[2]).

Contract (driver): `python bench.py --gpus N --steps K --warmup W`; for N>1 it is launched through
`torch.distributed.run` (one rank per GPU, RCCL) -- or spawns those ranks itself when WORLD_SIZE is unset.
At 1 GPU the same invocation also times vit_base (B=256, the other half of the headline metric), swin_t (B=128) and
alexnet (B=256, configs[0]'s model) and reports them under "extra".  A step = one forward of the per-GPU batch through the
HIP path (hipGraph replay of the recorded launch list) + the all-gather of the fp32 logits; images are
already resident in HBM.  W untimed warm-up steps, then exactly K timed steps bracketed by
barrier + synchronize; MAX over ranks; rank 0 prints ONE JSON line.

Extra objects in the line:
  roofline     the dominant kernel = the GEMM-type kernel (one name per kernel symbol) with the largest summed duration in the
               model traced as ONE launch list (the only setting in which per-kernel durations add up to a step and in which HIP
               events and rocprofv3 agree to 1-3 %).  `achieved` = its algorithmic FLOPs per launch (2 x MACs, `flop_per_launch`) /
               its average launch duration (`avg_launch_us`), measured live with HIP events on the launch stream right after the
               timed region; `rocprof` = the same kernel's `avg_us_warm` in the committed summary of `bench.py --lanes 1`
               (profiles/<round>/<model>_lanes1_rocprofv3_warm_stats.txt: the directory profiles/traffic.json names) and the fraction that follows from it.  `two_lanes`: the same
               measurement IN SITU in the two-lane graph's launch lists (eager two-stream replay; a launch then shares the chip with
               the other lane's kernel, and a profiler changes that overlap: profiles/r04/README.md, round 4's measurement);
               `frac` = achieved / 2.5 PFLOP/s (dense bf16 MFMA peak).  `rocprof`: the same kernel's warm average in the committed
               rocprofv3 summary (profiles/traffic.json -> profiles/<round>/<model>_rocprofv3_warm_stats.txt) and the fraction that
               follows from it.  `alone_frac` / `lanes1`: the same model as ONE launch list (lanes
               overlap in time, so only there does the sum over the dominant kernel's launches compare with a step);
               `isolated` (with --layers): every launch alone on the chip.  `whole_forward`: SURVEY section 8d's FLOPs per
               image x batch / the step (`frac`), algorithmic HBM bytes / the step / 8 TB/s (`hbm_frac`).
  cpu_baseline the CPU restatement (oracle/torch_ref.py, fp32, all host cores -- NOT JAX, which is not
               installed) timed on a bounded sample of the same workload, rank 0 at N=1 only.
`--layers FILE` additionally replays the recorded launch list kernel by kernel with HIP events and
writes a per-launch table (kernel, shape, us, TFLOP/s, GB/s) -- the tuning worksheet, outside the timing.
"""
from __future__ import annotations

import argparse
import ctypes
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _profile_dir() -> str:
    """The profiles/<round> directory that profiles/traffic.json (written by tools/collect_profiles.sh together with the files it
    cites) points at."""
    try:
        tj = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        for sect in ("_rocprof_lanes1", "_rocprof"):
            for mod in tj.get(sect, {}).values():
                for ent in mod.values():
                    return os.path.dirname(ent["file"])
    except Exception:  # noqa: BLE001
        pass
    return "profiles"
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import numpy as np  # noqa: E402
import torch  # noqa: E402

GFLOP_PER_IMG = {"resnet50": 8.178, "vit_base": 35.13, "swin_t": 8.98, "alexnet": 1.428,
                 "vgg16": 30.94, "vgg16_bn": 30.94, "vgg11": 15.22, "resnext50_32x4d": 8.46, "mobilenet_v2": 0.601, "mobilenet_v3_large": 0.434, "efficientnet_b0": 0.772, "efficientnet_v2_s": 16.8, "regnet_y_400mf": 0.80, "regnet_x_3_2gf": 6.35}   # section 8 f1 (2 x MACs of the conv / Linear layers)
MFMA_PEAK_TFLOPS = 2500.0     # dense bf16, /opt/skills/guides/MI355X_MICROARCH.md
HBM_PEAK_GBS = 8000.0
FORCE_COMM = os.environ.get("EQV_FORCE_COMM") == "1"     # 1 rank, logits still through mv_allgather (see run_model)


def build_model(name: str, seed: int = 1):
    import warnings

    import eqxvision_amd as eqv
    key = eqv.random.PRNGKey(seed)
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        if name == "resnet50":
            net = eqv.utils.randomize_batchnorm(eqv.models.resnet50(key=key), seed)
        elif name == "vit_base":
            net = eqv.models.vit_base(num_classes=1000, key=key)
        elif name == "swin_t":
            net = eqv.models.swin_t(key=key)
        elif name == "alexnet":
            net = eqv.models.alexnet(key=key)
        elif name in ("efficientnet_b0", "efficientnet_v2_s", "regnet_y_400mf", "regnet_x_3_2gf"):
            net = eqv.utils.randomize_batchnorm(getattr(eqv.models, name)(key=key), seed)
        elif name == "mobilenet_v3_large":
            net = eqv.utils.randomize_batchnorm(eqv.models.mobilenet_v3_large(key=key), seed)
        elif name == "mobilenet_v2":
            net = eqv.utils.randomize_batchnorm(eqv.models.mobilenet_v2(key=key), seed)
        elif name == "resnext50_32x4d":
            net = eqv.utils.randomize_batchnorm(eqv.models.resnext50_32x4d(key=key), seed)
        elif name in ("vgg16", "vgg16_bn", "vgg11"):
            net = getattr(eqv.models, name)(key=key)
            if name.endswith("_bn"):
                net = eqv.utils.randomize_batchnorm(net, seed)
        else:
            raise SystemExit(f"unknown model {name}")
    return eqv.tree_inference(net, True)


def _ev():
    from eqxvision_amd import _lib
    e = ctypes.c_void_p()
    _lib.call("mv_event_create", ctypes.byref(e))
    return e


def launch_meta(name, args):
    """(algorithmic FLOPs, algorithmic HBM bytes, shape string) of one recorded C-ABI call: 2 x MACs of the contraction, and the
    bytes the launch has to move at least once (operands + result + residual; fp32 tensors at 4 bytes)."""
    flops = byts = 0.0
    shape = ""
    ob = lambda dt: 4.0 if dt == 0 else 2.0                   # bytes per element of an MV_F32 / MV_BF16 tensor
    if name == "mv_conv2d_nhwc_fwd":
        N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, g = args[6:20]
        idt, odt = args[21], args[22]
        Ho = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
        flops = 2.0 * N * Ho * Wo * K * R * S * C / g
        byts = ob(idt) * (N * H * W * C + K * R * S * C / g) + ob(odt) * N * Ho * Wo * K * (2 if args[4] else 1)
        shape = f"N{N} {H}x{W}x{C}->{Ho}x{Wo}x{K} k{R}s{sh}"
    elif name == "mv_dwconv2d_nhwc_fwd":
        N, H, W, C, R, S, sh, sw, ph, pw, dh, dw = args[5:17]
        Ho = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
        flops = 2.0 * N * Ho * Wo * C * R * S
        byts = 2.0 * N * C * (H * W + Ho * Wo)
        shape = f"N{N} {H}x{W}x{C} dw k{R}s{sh}"
    elif name == "mv_conv2d_nhwc_grouped64_fwd":
        N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw, g = args[6:20]
        Ho = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
        Wo = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
        flops = 2.0 * N * Ho * Wo * K * R * S * C / g                  # the grouped convolution's own FLOPs (not the padded tile's)
        byts = 2.0 * (N * H * W * C + K * R * S * 64 + N * Ho * Wo * K)
        shape = f"N{N} {H}x{W}x{C}->{Ho}x{Wo}x{K} k{R}s{sh} g{g}"
    elif name in ("mv_linear_fwd", "mv_linear_split_fwd"):
        M, N, K = args[6:9]
        idt, odt = args[10], args[11]
        split = 2 if name == "mv_linear_split_fwd" else 1
        flops = 2.0 * M * N * K * split
        byts = ob(idt) * (M * K + split * N * K) + ob(odt) * M * N * (2 if args[4] else 1)
        shape = f"M{M} K{K} N{N}" + (" f32out" if odt == 0 else "") + (" +res" if args[4] else "") + (" hi+lo" if split == 2 else "")
    elif name == "mv_ln_linear_fwd":
        M, N, K = args[4:7]
        flops = 2.0 * M * N * K
        byts = ob(args[9]) * M * K + 2.0 * N * K + 2.0 * M * N
        shape = f"M{M} K{K} N{N} LN-in" + (" f32in" if args[9] == 0 else "")
    elif name == "mv_ln_mlp_fwd":
        M, C, Hd = args[6:9]
        flops = 4.0 * M * C * Hd
        byts = 2.0 * ob(args[10]) * M * C + 4.0 * C * Hd
        shape = f"M{M} C{C} hidden{Hd} LN+MLP+res"
    elif name == "mv_fc_stream_fwd":
        M, N, K = args[6:9]
        flops = 2.0 * M * N * K
        byts = 2.0 * (M * K + N * K) + ob(args[11]) * M * N
        shape = f"M{M} K{K} N{N} fc (fragment-ordered weights, split-K)" + (" f32out" if args[11] == 0 else "")
    elif name == "mv_ln_mlp_stream_fwd":
        M, C, Hd = args[6:9]
        flops = 4.0 * M * C * Hd
        byts = 2.0 * ob(args[10]) * M * C + 4.0 * C * Hd
        shape = f"M{M} C{C} hidden{Hd} LN+MLP+res (streamed weights)"
    elif name == "mv_swin_block_attn_fwd":
        B_, Hf, Wf, C, heads, wsh, wsw = args[7:14]
        M = B_ * Hf * Wf
        flops = M * (8.0 * C * C + 4.0 * wsh * wsw * C)            # qkv + proj Linears, Q.K^T and P.V inside a window
        byts = 2.0 * ob(args[17]) * M * C + 2.0 * 4 * C * C
        shape = f"B{B_} {Hf}x{Wf}x{C} h{heads} w{wsh} LN+qkv+attn+proj+res"
    elif name == "mv_patch_merge_ln_fwd":
        B_, H, W, C = args[4:8]
        byts = ob(args[9]) * B_ * H * W * C + ob(args[10]) * B_ * H * W * C
        shape = f"B{B_} {H}x{W}x{C} -> {H // 2}x{W // 2}x{4 * C} gather+LN"
    elif name == "mv_patch4_ln_fwd":
        B_, C, H, W, K = args[7:12]
        M = B_ * (H // 4) * (W // 4)
        flops = 2.0 * M * 16 * C * K * (2 if args[2] else 1)
        byts = 4.0 * B_ * C * H * W + 4.0 * M * K
        shape = f"B{B_} {C}x{H}x{W} -> {H // 4}x{W // 4}x{K} conv4s4+LN" + (" hi+lo" if args[2] else "")
    elif name == "mv_linear_lnout_fwd":      # residual stream as two bf16 planes (4 bytes per value, as fp32) + the row-statistics pieces
        M, N, K = args[8:11]
        flops = 2.0 * M * N * K
        byts = 2.0 * (M * K + N * K) + 8.0 * M * N + (8.0 * ((N + 255) // 256) * M if args[7] else 0.0)
        shape = f"M{M} K{K} N{N} +res " + ("f32 rows" if not args[4] else "planes") + " -> " + ("planes + row stats" if args[7] else "f32 rows")
    elif name == "mv_linear_lnin_fwd":
        M, N, K = args[6:9]
        flops = 2.0 * M * N * K
        byts = 2.0 * (M * K + N * K + M * N) + 8.0 * ((K + 255) // 256) * M
        shape = f"M{M} K{K} N{N} LN in the epilogue" + (" head-major" if args[11] else "")
    elif name == "mv_linear_heads_fwd":
        M, N, K = args[5:8]
        flops = 2.0 * M * N * K
        byts = 2.0 * (M * K + N * K + M * N)
        shape = f"M{M} K{K} N{N} head-major"
    elif name == "mv_conv1x1_dual_fwd":
        N, Ho, Wo, C1, H2, W2, C2, s2, K = args[6:15]
        M = N * Ho * Wo
        flops = 2.0 * M * K * (C1 + C2)
        byts = 2.0 * (M * C1 + M * C2 + K * (C1 + C2) + M * K)
        shape = f"M{M} {C1}+{C2}(s{s2})->{K}"
    elif name == "mv_bottleneck_tail_fwd":
        B_, H, W, wid, cout = args[9:14]
        M = B_ * H * W
        flops = 2.0 * M * wid * (9 * wid + cout)
        byts = 2.0 * (M * (wid + 2 * cout) + 9 * wid * wid + wid * cout)
        shape = f"N{B_} {H}x{W}x{wid} 3x3->{wid}->1x1->{cout}(+res)"
    elif name == "mv_conv1x1_chain_fwd":
        M, C, K, N2 = args[10:14]
        flops = 2.0 * M * (C * K + K * N2)
        byts = 2.0 * M * (C + 2 * K + N2)
        shape = f"M{M} {C}->{K}(+res)->{N2}"
    elif name == "mv_conv1x1_chain_rc0_fwd":        # the reference's work of this boundary (conv3 + downsample conv + next conv1); y is not stored
        M, C, K, N2 = args[5:9]
        flops = 2.0 * M * (2 * C * K + K * N2)
        byts = 2.0 * M * (2 * C + N2)
        shape = f"M{M} {C}+{C}->{K}(not stored)->{N2}"
    elif name == "mv_conv1x1_chain_rc_fwd":         # algorithmic FLOPs = conv3 + next conv1; the recomputed identity (2 M 2C K) is not counted
        M, C, K, N2 = args[7:11]
        flops = 2.0 * M * (C * K + K * N2)
        byts = 2.0 * M * (3 * C + K + N2)
        shape = f"M{M} {C}->{K}(+recomputed identity from {C}+{C})->{N2}"
    elif name == "mv_conv1x1_chain_res_fwd":
        N, H, W, C, K, N2, sub = args[6:13]
        M = N * H * W
        flops = 2.0 * M * (C * K + K * N2)
        byts = 2.0 * M * (C + K + (K // 4 if sub else K) + N2)
        shape = f"M{M} {C}->{K}(+res{', stored at even pixels' if sub else ''})->{N2}"
    elif name == "mv_conv1x1_chain_sub_fwd":
        N, H, W, C, K, N2 = args[10:16]
        M = N * H * W
        flops = 2.0 * M * (C * K + K * N2)
        byts = 2.0 * M * (C + K + K // 4 + N2)
        shape = f"M{M} {C}->{K}(+res, stored at even pixels)->{N2}"
    elif name == "mv_conv1x1_dual_chain_fwd":
        M, C1, C2, K, N2 = args[10:15]
        flops = 2.0 * M * ((C1 + C2) * K + K * N2)
        byts = 2.0 * M * (C1 + C2 + K + N2)
        shape = f"M{M} {C1}+{C2}->{K}->{N2}"
    elif name == "mv_stem_conv_pool_fwd":
        N, C, H, W, K, R, S, sh, sw, ph, pw, pk, ps, pp = args[5:19]
        Ho, Wo = (H + 2 * ph - R) // sh + 1, (W + 2 * pw - S) // sw + 1
        Po, Qo = (Ho + 2 * pp - pk) // ps + 1, (Wo + 2 * pp - pk) // ps + 1
        flops = 2.0 * N * Ho * Wo * K * R * S * C
        byts = ob(args[20]) * N * C * H * W + 2.0 * N * Po * Qo * K
        shape = f"N{N} {C}x{H}x{W}->conv{Ho}x{Wo}->pool{Po}x{Qo}x{K}"
    elif name in ("mv_conv2d_nchw_fwd", "mv_conv2d_nchw_split_fwd"):
        o = 1 if name.endswith("split_fwd") else 0
        N, C, H, W, K, R, S, sh, sw, ph, pw = args[5 + o:16 + o]
        Ho = (H + 2 * ph - R) // sh + 1
        Wo = (W + 2 * pw - S) // sw + 1
        flops = 2.0 * N * Ho * Wo * K * R * S * C * (1 + o)
        byts = 4.0 * N * C * H * W + 2.0 * N * Ho * Wo * K
        shape = f"N{N} {C}x{H}x{W}->{Ho}x{Wo}x{K} k{R}s{sh}" + (" hi+lo" if o else "")
    elif name in ("mv_mha_fwd", "mv_mha_heads_fwd"):
        B, N, H, dh = args[3:7]
        flops = 4.0 * B * H * N * N * dh
        byts = 2.0 * B * N * H * dh * 4
        shape = f"B{B} N{N} H{H} dh{dh}"
    elif name == "mv_swin_window_attn_fwd":
        B, Hf, Wf, C, heads, wh, ww = args[3:10]
        flops = 4.0 * B * Hf * Wf * (wh * ww) * C
        byts = 2.0 * B * Hf * Wf * C * 4
        shape = f"B{B} {Hf}x{Wf}x{C} h{heads} w{wh}"
    elif name == "mv_maxpool2d_nhwc_fwd":
        N, H, W, C, kh, kw, sh, sw, ph, pw = args[2:12]
        Ho = (H + 2 * ph - kh) // sh + 1
        byts = 2.0 * N * C * (H * W + Ho * Ho)
        shape = f"N{N} {H}x{W}x{C}"
    elif name == "mv_layernorm_fwd":
        M, C = args[4:6]
        byts = float(M) * C * (ob(args[8]) + ob(args[9]))
        shape = f"M{M} C{C}"
    return flops, byts, shape


def _symbol_name(kern, shape):
    """One table name per kernel SYMBOL (so a rocprofv3 row and a bench row are the same launches): a split-K launch runs the same
    igemm8s_kernel instance as the un-split one -- the suffix moves into the shape column."""
    if kern.endswith("_splitk"):
        return kern[:-7], shape + " split-K"
    return kern, shape


def layer_table(compiled, path, steps=5):
    """Replay the recorded launch list call by call with HIP events -> per-launch worksheet (every launch ALONE on the chip)."""
    from eqxvision_amd import _lib
    from eqxvision_amd._act import stream_ptr
    s = stream_ptr()
    rows = []
    e0, e1 = _ev(), _ev()
    for cfn, args, name in compiled.calls:
        cfn(*args[:-1], s)
        kern = _lib.last_kernel()
        ts = []
        for _ in range(steps):
            _lib.call("mv_event_record", e0, s)
            cfn(*args[:-1], s)
            _lib.call("mv_event_record", e1, s)
            ms = ctypes.c_float()
            _lib.call("mv_event_elapsed_ms", e0, e1, ctypes.byref(ms))
            ts.append(ms.value)
        us = 1e3 * float(np.median(ts))
        flops, byts, shape = launch_meta(name, args)
        kern, shape = _symbol_name(kern, shape)
        rows.append({"call": name, "kernel": kern, "shape": shape, "us": round(us, 2),
                     "tflops": round(flops / us / 1e6, 1) if us else 0, "gbs": round(byts / us / 1e3, 1) if us else 0,
                     "gflop": round(flops / 1e9, 3), "mb": round(byts / 1e6, 2)})
    tot = sum(r["us"] for r in rows)
    if path is None:
        return rows
    with open(path, "w") as f:
        f.write(f"# per-launch replay, {len(rows)} launches, sum {tot:.1f} us\n")
        f.write(f"{'kernel':34s} {'shape':38s} {'us':>9s} {'TFLOP/s':>8s} {'GB/s':>8s} {'GFLOP':>8s} {'MB':>8s} {'%':>5s}\n")
        for r in rows:
            f.write(f"{r['kernel']:34s} {r['shape']:38s} {r['us']:9.2f} {r['tflops']:8.1f} {r['gbs']:8.1f} "
                    f"{r['gflop']:8.3f} {r['mb']:8.2f} {100 * r['us'] / tot:5.1f}\n")
    return rows


def insitu_rows(compiled, steps=6, only=None):
    """Per-launch durations IN SITU: the recorded launch lists replayed eagerly the way the hipGraph runs them -- lane l on its own
    stream, forked and joined once per step -- with every launch bracketed by two HIP events on the stream it is launched on.
    This is the duration rocprofv3's kernel trace reports for the graph replays (two lanes share the chip, so a launch takes
    longer than alone); with one lane it is the plain back-to-back duration."""
    from eqxvision_amd import _lib
    lanes = compiled.lane_calls or [compiled.calls]
    main = torch.cuda.current_stream()
    streams = [torch.cuda.Stream() for _ in lanes]
    evs = [[(_ev(), _ev()) for _ in lc] for lc in lanes]
    kern = [[""] * len(lc) for lc in lanes]
    tot = [[0.0] * len(lc) for lc in lanes]
    n = max(len(lc) for lc in lanes)
    for st in range(steps + 1):                       # step 0 = warm-up (and kernel names)
        fork = torch.cuda.Event()
        fork.record(main)
        for s_ in streams:
            s_.wait_event(fork)
        for i in range(n):
            for l, lc in enumerate(lanes):
                if i < len(lc):
                    cfn, args, name = lc[i]
                    sp = streams[l].cuda_stream
                    timed = only is None or (l, i) in only
                    if timed:
                        _lib.call("mv_event_record", evs[l][i][0], sp)
                    rc = cfn(*args[:-1], sp)
                    if rc != 0:
                        raise RuntimeError(f"in-situ replay of {name} failed (rc={rc})")
                    if st == 0:
                        kern[l][i] = _lib.last_kernel()
                    if timed:
                        _lib.call("mv_event_record", evs[l][i][1], sp)
        for s_ in streams:
            d = torch.cuda.Event()
            d.record(s_)
            main.wait_event(d)
        torch.cuda.synchronize()
        if st == 0:
            continue
        for l, lc in enumerate(lanes):
            for i in range(len(lc)):
                if only is not None and (l, i) not in only:
                    continue
                ms = ctypes.c_float()
                _lib.call("mv_event_elapsed_ms", evs[l][i][0], evs[l][i][1], ctypes.byref(ms))
                tot[l][i] += ms.value
    rows = []
    for l, lc in enumerate(lanes):
        for i, (cfn, args, name) in enumerate(lc):
            flops, byts, shape = launch_meta(name, args)
            us = 1e3 * tot[l][i] / steps
            kern[l][i], shape = _symbol_name(kern[l][i], shape)
            rows.append({"call": name, "kernel": kern[l][i], "shape": shape, "us": round(us, 2), "lane": l, "index": i,
                         "tflops": round(flops / us / 1e6, 1) if us else 0, "gbs": round(byts / us / 1e3, 1) if us else 0,
                         "gflop": round(flops / 1e9, 3), "mb": round(byts / 1e6, 2)})
    return rows



def dominant_family(rows, name=None):
    """The GEMM-type kernel (one name = one kernel symbol, so a rocprofv3 summary row covers the same launches) with the largest
    summed duration in one forward -- or the kernel called `name` -- -> (name, {us, gflop, mb, n})."""
    fam = {}
    for r_ in rows:
        d = fam.setdefault(r_["kernel"], {"us": 0.0, "gflop": 0.0, "mb": 0.0, "n": 0})
        d["us"] += r_["us"]; d["gflop"] += r_["gflop"]; d["mb"] += r_["mb"]; d["n"] += 1
    if name is not None and name in fam:
        return name, fam[name]
    gemm = {k: v for k, v in fam.items() if v["gflop"] > 0}
    return max((gemm or fam or {"n/a": {"us": 1, "gflop": 0, "mb": 0, "n": 1}}).items(), key=lambda kv: kv[1]["us"])


def cpu_baseline(model_name, net, threads):
    """CPU restatement (port) of the same forward on the host cores; bounded sample."""
    import eqxvision_amd as eqv
    from oracle import torch_ref as TR
    if model_name == "swin_t":              # the restatement reads torchvision's key names; same architecture, synthetic checkpoint
        from oracle import state as S_
        sd = S_.swin_state(1)
    else:
        sd = eqv.utils.state_dict(net)
    B = 32 if model_name != "vit_base" else 16
    reps = 4 if model_name != "vit_base" else 3
    x = np.random.Generator(np.random.PCG64(0)).random((B, 3, 224, 224), dtype=np.float32)
    fwd = {"resnet50": lambda: TR.resnet_forward(sd, x), "vit_base": lambda: TR.vit_forward(sd, x),
           "swin_t": lambda: TR.swin_forward(sd, x), "alexnet": lambda: TR.alexnet_forward(sd, x)}.get(model_name)
    if fwd is None:
        return None
    t_all = time.time()
    best, used = 1e30, threads
    for nt in sorted({min(threads, 32), threads}):      # many-core hosts: oversubscription can be slower
        torch.set_num_threads(nt)
        fwd()                               # warm-up
        for _ in range(reps if nt == threads else 2):
            t = time.time()
            fwd()
            dt_ = time.time() - t
            if dt_ < best:
                best, used = dt_, nt
    threads = used
    return {"value": round(B / best, 2), "unit": "images/s", "cores": threads, "kind": "port",
            "sample": f"{model_name} fp32 forward (torch-CPU restatement oracle/torch_ref.py, NOT JAX), batch {B}, "
                      f"best of {reps} after 1 warm-up ({time.time() - t_all:.1f}s of CPU work)"}


def run_model(a, name, B, rank, world, soak_s):
    """Build `name`, trace + capture its forward, (soak), W warm-up steps, K timed steps -> result dict (rank 0) + net."""
    import eqxvision_amd as eqv
    from eqxvision_amd import _lib, dist as D
    from eqxvision_amd._act import stream_ptr

    net = build_model(name)
    # synthetic images, generated once, resident in HBM (fp32 NCHW like the reference's inputs)
    g = torch.Generator(device="cpu").manual_seed(rank)
    images = torch.rand((B, 3, 224, 224), generator=g, dtype=torch.float32).cuda()
    keys = eqv.random.split(eqv.random.PRNGKey(0), B)
    fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k),
                         use_graph=not a.no_graph, clone_outputs=False, lanes=a.lanes)
    # EQV_FORCE_COMM=1 (tests/test_dist_gpu.py): with ONE rank the logits still go through mv_allgather on a 1-rank RCCL
    # communicator inside the timed region -- the code path the 8-GPU line takes, executed on the hardware a 1-GPU box has
    use_comm = world > 1 or (FORCE_COMM and D._state["native"])
    gathered = torch.empty((B * world, 1000), dtype=torch.float32, device="cuda") if use_comm else None
    local_logits = [None]

    def step():
        logits = fwd(net, images, keys)
        if use_comm:                         # ONE all-gather of the fp32 logits on the launch stream (mv_allgather / RCCL)
            local_logits[0] = logits
            logits = D.all_gather_rows(logits, B * world, out=gathered)
        return logits

    for _ in range(3):                       # "compile": call 1 records, call 2 captures the hipGraph, call 3 = first replay
        out = step()
    torch.cuda.synchronize()
    if soak_s > 0:                           # untimed: bring clocks / thermals to the sustained state (and make the GPU
        t_end = time.perf_counter() + soak_s  # phase of this short benchmark visible to coarse utilisation samplers)
        while time.perf_counter() < t_end:
            for _ in range(20):
                step()
            torch.cuda.synchronize()
    for _ in range(a.warmup):                # the W untimed warm-up steps proper (graph replays)
        out = step()
    torch.cuda.synchronize()
    assert out.shape == (B * world, 1000) and bool(torch.isfinite(out).all())
    if use_comm and world == 1:              # forced 1-rank collective: the gathered buffer IS the local logits, bit for bit
        assert out.data_ptr() == gathered.data_ptr() and out.data_ptr() != local_logits[0].data_ptr()
        assert torch.equal(out, local_logits[0]), "mv_allgather on a 1-rank communicator changed the logits"

    e0, e1 = _ev(), _ev()
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    _lib.call("mv_event_record", e0, stream_ptr())
    for _ in range(a.steps):
        step()
    _lib.call("mv_event_record", e1, stream_ptr())
    torch.cuda.synchronize()
    D.barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    ms = ctypes.c_float()
    _lib.call("mv_event_elapsed_ms", e0, e1, ctypes.byref(ms))
    dev_ms_per_step = ms.value / a.steps
    dt = D.max_over_ranks(dt)
    compiled = fwd._entries()[0]
    if rank != 0:
        return None, net
    # ---- the same forward with the lanes NOT joined per call (filter_jit(join="stream"), an opt-in): reported beside the line, never
    # as `value` -- the returned logits are futures until `ready()`, a different contract from the default's
    pipelined = None
    if world == 1 and not a.no_graph and a.lanes >= 2 and not a.no_pipelined:
        try:
            fp = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), clone_outputs=False, lanes=a.lanes,
                                join="stream")
            for _ in range(6):               # two buffer sets: calls 1-2 record, 3-4 capture the per-lane graphs, 5-6 replay
                op = fp(net, images, keys)
            fp.block_until_ready()
            same = bool(torch.equal(op, out))
            torch.cuda.synchronize()
            tp0 = time.perf_counter()
            for _ in range(a.steps):
                op = fp(net, images, keys)
            fp.block_until_ready()
            torch.cuda.synchronize()
            dtp = time.perf_counter() - tp0
            pipelined = {"value": round(B * a.steps / dtp, 1), "unit": "images/s", "ms_per_step": round(1e3 * dtp / a.steps, 4),
                         "logits_bit_identical_to_the_joined_forward": same,
                         "how": f"filter_jit(lanes={a.lanes}, join='stream'): one hipGraph per lane on its own stream, no join per call "
                                f"(call i+1's first lane starts under call i's last); {a.steps} calls, then ready() + synchronize, "
                                "host clock.  Not the headline: the result of a call is a future until ready()."}
            del fp
        except Exception as e:  # noqa: BLE001
            pipelined = {"error": f"{type(e).__name__}: {e}"}
    step_ms = 1e3 * dt / a.steps
    flop_per_launch = GFLOP_PER_IMG[name] * 1e9 * B
    achieved = flop_per_launch / (dev_ms_per_step * 1e-3) / 1e12
    # ---- roofline of the dominant kernel, measured live with HIP events on the launch streams (docstring) -----------------------
    rows = insitu_rows(compiled)
    dk, dv = dominant_family(rows)
    dom_tflops = dv["gflop"] / dv["us"] * 1e3 if dv["us"] else 0.0   # GFLOP/us = PFLOP/s
    nl = len(compiled.lane_calls) if compiled.lane_calls else 1
    alg_mb = sum(r_["mb"] for r_ in rows)
    bound_us = sum(max(r_["gflop"] * 1e3 / MFMA_PEAK_TFLOPS, r_["mb"] / HBM_PEAK_GBS * 1e3) for r_ in rows)
    traffic, rocprof = None, None
    tj = os.path.join(ROOT, "profiles", "traffic.json")       # PMC-derived HBM bytes per launch (separate --pmc passes) and the
    if os.path.exists(tj) and world == 1:                      # committed rocprofv3 summary's duration of the same kernel, if collected
        try:
            tjd = json.load(open(tj))
            if tjd.get("_batch", {}).get(name) == B:
                traffic = tjd.get(name, {}).get(dk)
                rp = tjd.get("_rocprof", {}).get(name, {}).get(dk)
                if rp and dv["n"]:
                    fl = dv["gflop"] / dv["n"]              # GFLOP per launch
                    rocprof = {"file": rp.get("file"), "avg_us_warm": rp["avg_launch_us"], "calls": rp.get("calls"),
                               "frac": round(fl / rp["avg_launch_us"] * 1e3 / MFMA_PEAK_TFLOPS, 4),
                               "note": "the committed rocprofv3 --kernel-trace summary of this command (two lanes), same kernel symbol; "
                                       "frac = flop_per_launch / avg_us_warm / peak"}
        except Exception:  # noqa: BLE001
            traffic = None
    roof = {"bound": "mfma", "achieved": round(dom_tflops, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s",
            "frac": round(dom_tflops / MFMA_PEAK_TFLOPS, 4), "traffic": traffic, "kernel": dk, "launches_per_step": dv["n"],
            "rocprof": rocprof,
            "avg_launch_us": round(dv["us"] / max(1, dv["n"]), 2),
            "how": f"in situ: eager replay of the {nl}-lane launch lists on {nl} stream(s), two HIP events around every launch on "
                   "its own stream, 6 steps (the LAST 6 x launches_per_step rows of a rocprofv3 --kernel-trace of this command are "
                   f"this pass: {_profile_dir()}/roofline_vs_rocprof.txt compares the two clocks)",
            "share_of_kernel_time": round(dv["us"] / max(1e-9, sum(r_["us"] for r_ in rows)), 3),
            "flop_per_launch": round(dv["gflop"] * 1e9 / max(1, dv["n"])),
            "kernel_hbm_gbs": round(dv["mb"] / dv["us"] * 1e3, 1) if dv["us"] else 0.0,
            "whole_forward": {"achieved": round(achieved, 1), "frac": round(achieved / MFMA_PEAK_TFLOPS, 4),
                              "graph_launch_ms": round(dev_ms_per_step, 4), "flop_per_launch": flop_per_launch,
                              # algorithmic HBM bytes of one forward (sum over the launches; fused launches at what they have to
                              # move) over the measured step, against 8 TB/s
                              "algorithmic_mb": round(alg_mb, 1),
                              "hbm_frac": round(alg_mb * 1e6 / (step_ms * 1e-3) / (HBM_PEAK_GBS * 1e9), 4),
                              # layer-wise speed of light: sum over launches of max(flops / MFMA peak, algorithmic bytes /
                              # HBM peak) over the measured step (1.0 = every launch on its own roofline, nothing overlapped)
                              "layerwise_bound_ms": round(bound_us / 1e3, 4),
                              "layerwise_bound_frac": round(bound_us / 1e3 / dev_ms_per_step, 4)}}
    if nl > 1 and not a.no_lanes1:
        # lanes overlap in time, so the in-situ durations of the two lanes do not add up to the step; the same model traced as ONE
        # launch list (full-batch launches back to back on one stream) gives per-kernel figures whose sum is comparable with a step
        f1 = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=False, clone_outputs=False, lanes=1)
        for _ in range(2):
            f1(net, images, keys)
        torch.cuda.synchronize()
        e0_, e1_ = _ev(), _ev()
        _lib.call("mv_event_record", e0_, stream_ptr())
        for _ in range(5):
            f1(net, images, keys)
        _lib.call("mv_event_record", e1_, stream_ptr())
        torch.cuda.synchronize()
        ms1 = ctypes.c_float()
        _lib.call("mv_event_elapsed_ms", e0_, e1_, ctypes.byref(ms1))
        rows1 = insitu_rows(f1._entries()[0])
        k1, v1 = dominant_family(rows1)
        t1 = v1["gflop"] / v1["us"] * 1e3 if v1["us"] else 0.0
        two = {k: roof[k] for k in ("kernel", "achieved", "frac", "launches_per_step", "avg_launch_us", "how", "share_of_kernel_time",
                                     "flop_per_launch", "kernel_hbm_gbs", "rocprof")}
        # THE headline figure: the dominant kernel of the ONE-lane list.  Only there do the per-kernel durations add up to a step
        # (sum_all_kernels_ms ~ ms_per_step_eager) and only there do HIP events and a rocprofv3 trace of `bench.py --lanes 1` agree
        # (1-3 %): with two lanes a launch shares the chip with the other lane's kernel, and rocprofv3 itself changes how much the
        # lanes overlap (round 4's measurement: profiles/r04/README.md), so the two-lane in-situ figure (`roofline.two_lanes`) has no second clock.
        rp1 = None
        try:
            tjd = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
            rp = tjd.get("_rocprof_lanes1", {}).get(name, {}).get(k1) if tjd.get("_batch", {}).get(name) == B and world == 1 else None
            if rp and v1["n"]:
                rp1 = {"file": rp.get("file"), "avg_us_warm": rp["avg_launch_us"], "calls": rp.get("calls"),
                       "frac": round(v1["gflop"] / v1["n"] / rp["avg_launch_us"] * 1e3 / MFMA_PEAK_TFLOPS, 4),
                       "note": "committed rocprofv3 --kernel-trace summary of `bench.py --lanes 1` (same kernel symbol): "
                               "frac = flop_per_launch / avg_us_warm / peak"}
        except Exception:  # noqa: BLE001
            rp1 = None
        roof.update({"kernel": k1, "achieved": round(t1, 1), "frac": round(t1 / MFMA_PEAK_TFLOPS, 4), "alone_frac": round(t1 / MFMA_PEAK_TFLOPS, 4),
                     "launches_per_step": v1["n"], "avg_launch_us": round(v1["us"] / max(1, v1["n"]), 2),
                     "flop_per_launch": round(v1["gflop"] * 1e9 / max(1, v1["n"])),
                     "kernel_hbm_gbs": round(v1["mb"] / v1["us"] * 1e3, 1) if v1["us"] else 0.0,
                     "share_of_kernel_time": round(v1["us"] / max(1e-9, sum(r_["us"] for r_ in rows1)), 3),
                     "how": "one lane: the model traced as ONE launch list (full-batch launches back to back on one stream), two HIP "
                            "events around every launch, 6 eager replays; the dominant kernel = largest summed duration among the "
                            "GEMM-type kernels (one name per kernel symbol)",
                     "rocprof": rp1, "two_lanes": two})
        # `traffic` belongs to the headline kernel: PMC bytes per launch of k1.  The PMC passes ran the two-lane command (launches of
        # B / nl images, profiles/traffic.json); a one-lane launch carries the whole batch, so the per-launch figure is scaled by nl
        # (the kernels' traffic is linear in the batch: weights are a few MB of 100+).
        try:
            t_k1 = tjd.get(name, {}).get(k1) if tjd.get("_batch", {}).get(name) == B and world == 1 else None
        except Exception:  # noqa: BLE001
            t_k1 = None
        roof["traffic"] = None if t_k1 is None else int(t_k1 * nl)
        roof["traffic_how"] = (None if t_k1 is None else
                               f"rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE (separate passes) per launch of {k1} in the two-lane command "
                               f"({B // nl} images per launch: {t_k1} bytes), x {nl} for this full-batch launch; profiles/traffic.json")
        two["traffic"] = traffic if traffic is not None and two["kernel"] == dk else None
        # both fractions side by side under explicit names (advisor, round 4): `frac` / `frac_lanes1` = the kernel alone on the chip
        # (one launch list), `frac_in_situ` = the same-symbol or dominant two-lane kernel while the other lane runs -- the
        # configuration `value` is measured in; `whole_forward.frac` is the time-weighted figure of the whole step
        roof["frac_lanes1"] = roof["frac"]
        roof["frac_in_situ"] = two["frac"]
        roof["lanes1"] = {"kernel": k1, "launches_per_step": v1["n"], "avg_launch_us": round(v1["us"] / max(1, v1["n"]), 2),
                          "achieved": round(t1, 1), "frac": round(t1 / MFMA_PEAK_TFLOPS, 4),
                          "sum_dominant_ms": round(v1["us"] / 1e3, 4), "sum_all_kernels_ms": round(sum(r_["us"] for r_ in rows1) / 1e3, 4),
                          "ms_per_step_eager": round(ms1.value / 5, 4)}
        f1._cache.clear()
    if a.layers and name == a.model:                       # tuning worksheet: every launch ALONE on the chip
        iso = layer_table(compiled, a.layers)
        ik, iv = dominant_family(iso)
        roof["isolated"] = {"kernel": ik, "avg_launch_us": round(iv["us"] / max(1, iv["n"]), 2),
                            "frac": round(iv["gflop"] / iv["us"] * 1e3 / MFMA_PEAK_TFLOPS, 4) if iv["us"] else 0.0}
    res = {"value": round(B * world * a.steps / dt, 1), "ms_per_step": round(step_ms, 4),
           "config": {"workload": f"{name} {a.dtype} forward, batch={B}/GPU, 3x224x224, 1000 classes",
                      "global_batch": B * world, "parallelism": f"dp{world}", "launches_per_step": len(compiled.calls),
                      "graph": compiled.graph is not None,
                      "lanes": len(compiled.lane_calls) if compiled.lane_calls else 1,
                      "collective": ("mv_allgather (RCCL)" if D._state["native"] else "torch.distributed") if use_comm else None,
                      "collective_forced_1rank": bool(use_comm and world == 1)},
           "roofline": roof}
    if pipelined is not None:
        res["lanes_not_joined"] = pipelined
    return res, net


def respawn(a):
    """`python bench.py --gpus N` without a launcher: re-exec under torch.distributed.run, one rank per GPU."""
    import socket
    import subprocess
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--model", default="resnet50")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default 256; 128 for swin_t)")
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--layers", default=None, help="write a per-launch worksheet of --model to this file")
    ap.add_argument("--no-pipelined", action="store_true", help="skip the join='stream' (lanes not joined per call) measurement")
    ap.add_argument("--extra", default=None,
                    help="comma list of further models timed in the same invocation and reported under \"extra\" "
                         "(default at 1 GPU for the headline model: vit_base,swin_t,alexnet; 'none' = only --model)")
    ap.add_argument("--soak", type=float, default=None, help="seconds of untimed graph replays before the warm-up steps "
                                                             "(default 5 at 1 GPU, 2 otherwise)")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-lanes1", action="store_true", help="skip the single-launch-list pass behind roofline.lanes1 (profile runs: "
                                                             "keeps full-batch launches of the same kernels out of the trace)")
    ap.add_argument("--lanes", type=int, default=2,
                    help="sub-batches captured as parallel hipGraph branches (fills the last, partial round of CUs of "
                         "one kernel with the other lane's next kernel); 1 = a single launch list")
    a = ap.parse_args()
    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(respawn(a))

    import eqxvision_amd as eqv
    from eqxvision_amd import dist as D

    rank, world, local = D.init_from_env()
    if world != a.gpus:
        raise SystemExit(f"--gpus {a.gpus} but {world} rank(s) joined (WORLD_SIZE={os.environ.get('WORLD_SIZE')}): "
                         "launch through torch.distributed.run or let bench.py spawn the ranks itself")
    if world > 1 and not D._state["native"] and D._state["collective"] == "rccl":
        raise SystemExit("the logits all-gather is NOT on mv_allgather (RCCL): mv_comm_init failed and the run fell back to "
                         "torch.distributed -- refusing to print a line (set EQV_DIST_COLLECTIVE=torch to measure that path on purpose)")
    if world > 1 and D._state["native"] and eqv._lib.load().mv_comm_size() != world:
        raise SystemExit(f"RCCL communicator has {eqv._lib.load().mv_comm_size()} ranks, expected {world}")
    torch.cuda.set_device(local)
    if FORCE_COMM and world == 1:
        D.native_comm_init(0, 1)             # raises if librccl cannot give a 1-rank communicator: no silent skip
    eqv.set_compute_dtype(a.dtype)
    default_batch = {"swin_t": 128}
    extras = []
    if world == 1 and a.extra != "none":
        extras = [m for m in (a.extra.split(",") if a.extra else (["vit_base", "swin_t", "alexnet"] if a.model == "resnet50" else [])) if m]
    soak = a.soak if a.soak is not None else (5.0 if world == 1 else 2.0)

    # CPU baseline FIRST (rank 0, 1 GPU only), so that the GPU phase is the tail of the run
    cpu = {}
    if world == 1 and not a.no_cpu:
        for m in [a.model] + [e for e in extras if e in ("vit_base", "swin_t", "alexnet")]:
            try:
                cpu[m] = cpu_baseline(m, build_model(m), torch.get_num_threads())
            except Exception as e:  # noqa: BLE001
                cpu[m] = {"error": f"{type(e).__name__}: {e}"}

    res, _ = run_model(a, a.model, a.batch or default_batch.get(a.model, 256), rank, world, soak)
    extra_out = {}
    for m in extras:
        try:
            r, _ = run_model(a, m, default_batch.get(m, 256), rank, world, min(soak, 3.0))
            if m in cpu:
                r["cpu_baseline"] = cpu[m]
            extra_out[m] = r
        except Exception as e:  # noqa: BLE001
            extra_out[m] = {"error": f"{type(e).__name__}: {e}"}
    if rank == 0:
        line = {"metric": "images/sec", "value": res["value"], "unit": "images/s", "n_gpus": world, "steps": a.steps,
                "warmup": a.warmup, "ms_per_step": res["ms_per_step"], "higher_is_better": True, "scaling": "weak",
                "vs_baseline": None, "dtype": a.dtype, "data": "synthetic", "config": res["config"], "roofline": res["roofline"]}
        if "lanes_not_joined" in res:        # opt-in filter_jit(join="stream"): beside the line, never its `value`
            line["lanes_not_joined"] = res["lanes_not_joined"]
        if a.model in cpu:
            line["cpu_baseline"] = cpu[a.model]
        if extra_out:
            line["extra"] = extra_out
        print(json.dumps(line), flush=True)
    if world > 1:
        import torch.distributed as td
        D.native_comm_destroy()
        td.destroy_process_group()
    elif FORCE_COMM:
        D.native_comm_destroy()


if __name__ == "__main__":
    main()
