"""Greedy per-shape kernel search on WHOLE-MODEL time (one process, one GPU box; box-to-box variance is larger than the
effects looked for).  For every GEMM-shaped launch of the model's forward that the igemm / stream kernels serve, try
the alternative kernels through the "ov:..." override flags (igemm.hip: tile_override) and keep a choice only when the
whole forward gets faster by more than THRESH.  Prints rows for eqxvision_amd/csrc/tuned_tiles.h.

usage: tune_tiles.py MODEL [BATCH] [LANES]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from eqxvision_amd import _lib
from bench import build_model

model = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
LANES = int(sys.argv[3]) if len(sys.argv) > 3 else 2
THRESH = float(os.environ.get("THRESH", "0.004"))
STEPS = int(os.environ.get("STEPS", "30"))
NAMES = {1: "igemm2 256x64", 2: "igemm2 256x128", 3: "igemm2 256x256", 7: "igemm 128x128", 8: "igemm 128x64", 9: "stream1x1",
         10: "igemm8 256x256", 11: "igemm8 128x256", 12: "igemm8 256x128"}

eqv.set_compute_dtype("bf16")
net = build_model(model)
images = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)


def make():
    f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False, lanes=LANES)
    for _ in range(3):
        f(net, images, keys)
    torch.cuda.synchronize()
    return f


def timeit(f, reps=2):
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(STEPS):
            f(net, images, keys)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / STEPS)
    return best


def shapes_of(f):
    """(kind, key tuple, default kernel, us) per distinct GEMM-shaped launch, from the recorded launch list."""
    c = f._entries()[0]
    calls = c.lane_calls[0] if c.lane_calls else c.calls
    s = torch.cuda.current_stream().cuda_stream
    out = {}
    for cfn, args, name in calls:
        if name == "mv_conv2d_nhwc_fwd":
            N, H, W, C, K, R, S, sh, sw, ph, pw, dh, dw = args[6:19]
            Ho = (H + 2 * ph - dh * (R - 1) - 1) // sh + 1
            Wo = (W + 2 * pw - dw * (S - 1) - 1) // sw + 1
            key = ("ov", N * Ho * Wo, C, K, R, S, sh)
        elif name == "mv_linear_fwd":
            M, N_, K_ = args[6:9]
            key = ("ov", M, K_, N_, 1, 1, 1)
        elif name == "mv_linear_heads_fwd":
            M, N_, K_ = args[5:8]
            key = ("ovh", M, N_, K_, 1, 1, 1)
        elif name == "mv_conv1x1_dual_fwd":
            N, Ho, Wo, C1, H2, W2, C2, s2, K_ = args[6:15]
            key = ("ovd", N * Ho * Wo, C1, C2, K_, s2, 1)
        else:
            continue
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        cfn(*args[:-1], s)
        kern = _lib.last_kernel()
        torch.cuda.synchronize()
        e0.record()
        for _ in range(3):
            cfn(*args[:-1], s)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 3 * 1e3
        o = out.setdefault(key, [kern, 0.0, 0])
        o[1] += us
        o[2] += 1
    return out


def flag_name(key):
    return ":".join(str(v) for v in key)


f0 = make()
sh = shapes_of(f0)
tunable = {k: v for k, v in sh.items() if v[0].startswith(("igemm", "stream1x1"))}
order = sorted(tunable, key=lambda k: -tunable[k][1])
print(f"# {model} B={B} lanes={LANES}: {len(order)} tunable shapes", flush=True)
base = timeit(f0)
print(f"# baseline {base:.4f} ms/step", flush=True)
chosen = {}
for key in order:
    kern, us, n = tunable[key]
    dense = key[0] == "ovh" or (key[4] == 1 and key[5] == 1 and key[6] == 1)
    if key[0] == "ovd":
        cands = [3, 10, 11]
    elif os.environ.get("CANDS"):
        cands = [int(c) for c in os.environ["CANDS"].split(",")]
    else:
        cands = [2, 3, 10, 11] if key[0] == "ovh" else ([2, 3, 7, 9, 10, 11, 12] if dense else [2, 3, 7, 10, 11, 12])
    cur = timeit(make())                      # the configuration so far, re-measured next to its challengers
    best_c, best_t = 0, cur
    log = []
    for cnd in cands:
        _lib.set_flag(flag_name(key), cnd)
        try:
            t = timeit(make())
        except Exception as e:                # a kernel that cannot express the shape
            t = 1e9
        log.append(f"{cnd}:{t:.4f}")
        if t < best_t * (1 - THRESH):
            best_c, best_t = cnd, t
    _lib.set_flag(flag_name(key), best_c)
    if best_c:
        chosen[key] = best_c
    print(f"# {flag_name(key):34s} default {kern:28s} {us:7.1f} us x{n}: now {cur:.4f} | " + " ".join(log) +
          (f"  -> {best_c} ({NAMES[best_c]}) {best_t:.4f}" if best_c else "  -> rules"), flush=True)
final = timeit(make(), reps=3)
print(f"# final {final:.4f} ms/step vs baseline {base:.4f} ({100 * (base / final - 1):+.2f}%)")
for key, c in chosen.items():
    print(f"TUNED {key[0]} {{{key[1]}, {key[2]}, {key[3]}, {key[4]}, {key[5]}, {key[6]}, {c}}},   // {model} B={B} lanes={LANES}: {NAMES[c]}")
