"""A/B of the GEMM cores on the hot shapes, one process, interleaved: default dispatch vs `igemm8=2` (tools only).
usage: g8_bench.py [lin|conv|all] [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from eqxvision_amd import _lib as L

S = lambda: torch.cuda.current_stream().cuda_stream


def timed(go, n=10):
    for _ in range(2):
        go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n):
        go()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def lin(M, N, K, act=0, res=False, f32=False):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=odt)

    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), M, N, K, act, 1, 0 if f32 else 1, S())
    return go, 2.0 * M * N * K, y


def conv(N, H, C, K, R, stride=1, res=False):
    pad = R // 2
    x = torch.randn(N, H, H, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(K, R, R, C, device="cuda") / (C * R * R) ** 0.5).to(torch.bfloat16)
    sc = torch.rand(K, device="cuda") + 0.5
    sh = torch.randn(K, device="cuda") * 0.1
    Ho = (H + 2 * pad - R) // stride + 1
    r = torch.randn(N, Ho, Ho, K, device="cuda").to(torch.bfloat16) if res else None
    y = torch.empty(N, Ho, Ho, K, device="cuda", dtype=torch.bfloat16)

    def go():
        L.call("mv_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), N, H, H, C, K, R, R, stride, stride, pad, pad, 1, 1, 1, 1, 1, 1, S())
    return go, 2.0 * N * Ho * Ho * K * R * R * C, y


def dual(N, Ho, C1, C2, K, s2):
    H2 = Ho * s2
    x = torch.randn(N, Ho, Ho, C1, device="cuda").to(torch.bfloat16)
    x2 = torch.randn(N, H2, H2, C2, device="cuda").to(torch.bfloat16)
    w = (torch.randn(K, C1 + C2, device="cuda") / (C1 + C2) ** 0.5).to(torch.bfloat16)
    sh = torch.randn(K, device="cuda") * 0.1
    y = torch.empty(N, Ho, Ho, K, device="cuda", dtype=torch.bfloat16)

    def go():
        L.call("mv_conv1x1_dual_fwd", x.data_ptr(), x2.data_ptr(), w.data_ptr(), None, sh.data_ptr(), y.data_ptr(), N, Ho, Ho, C1,
               H2, H2, C2, s2, K, 1, 1, S())
    return go, 2.0 * N * Ho * Ho * K * (C1 + C2), y


def ab(name, mk, reps, variants):
    go, flops, y = mk()
    res = {v: [] for v, _ in variants}
    kern = {}
    outs = {}
    for _ in range(reps):
        for v, flags in variants:
            for f, val in flags:
                L.set_flag(f, val)
            try:
                us = timed(go)
                kern[v] = L.last_kernel()
                outs[v] = y.float().clone()
            except Exception as e:  # noqa: BLE001
                us = float("nan")
                kern[v] = f"ERR {e}"[:60]
            for f, val in flags:
                L.set_flag(f, 0)
            res[v].append(us)
    base = outs.get(variants[0][0])
    line = f"{name:34s}"
    for v, _ in variants:
        us = float(np.median(res[v]))
        d = float((outs[v] - base).abs().max()) if v in outs and base is not None else float("nan")
        line += f" | {v}:{kern[v][7:26]:19s} {us:7.1f}us {flops / us / 1e6:6.0f}TF d={d:.2g}"
    print(line, flush=True)


def main():
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    variants = [("old", (("no_igemm8", 1),)), ("rule", ()), ("g8", (("igemm8", 2),)), ("s128x256", (("igemm8", 3),)),
                ("s256x128", (("igemm8", 4),))]
    extra = os.environ.get("G8_EXTRA")          # e.g. "relax:flag=1+flag2=3,other:flag=2"
    if extra:
        for item in extra.split(","):
            n, fvs = item.split(":")
            variants.append((n, tuple((fv.split("=")[0], int(fv.split("=")[1])) for fv in fvs.split("+"))))
    if what in ("lin", "all"):
        ab("8192^3", lambda: lin(8192, 8192, 8192), reps, variants)
        ab("4096^3", lambda: lin(4096, 4096, 4096), reps, variants)
        for B in (256, 128):
            M = 197 * B
            ab(f"vit qkv M{M}", lambda: lin(M, 2304, 768), reps, variants)
            ab(f"vit proj f32res M{M}", lambda: lin(M, 768, 768, res=True, f32=True), reps, variants)
            ab(f"vit fc1 gelu M{M}", lambda: lin(M, 3072, 768, act=2), reps, variants)
            ab(f"vit fc2 f32res M{M}", lambda: lin(M, 768, 3072, res=True, f32=True), reps, variants)
    if what in ("swin", "all"):
        for B in (128, 64):
            for (tok, Nn, Kk, res) in ((3136, 96, 384, True), (784, 192, 384, False), (784, 192, 768, True), (196, 1152, 384, False),
                                       (196, 384, 384, True), (196, 1536, 384, False), (196, 384, 1536, True), (49, 768, 1536, False),
                                       (49, 2304, 768, False), (49, 768, 768, True), (49, 3072, 768, False), (49, 768, 3072, True)):
                ab(f"swin M{tok * B} N{Nn} K{Kk}" + (" f32res" if res else ""), lambda: lin(tok * B, Nn, Kk, res=res, f32=res), reps,
                   variants)
    if what in ("dual", "all"):
        dv = [("base", (("ovd:%d", 3),)), ("g8", (("igemm8", 2),))]
        for B in (256, 128):
            for (Ho, C1, C2, K) in ((28, 128, 256, 512), (14, 256, 512, 1024), (7, 512, 1024, 2048)):
                M = B * Ho * Ho
                v = [("old", ((f"ovd:{M}:{C1}:{C2}:{K}:2:1", 3),)), ("rule", ()), ("g8", (("igemm8", 2),)), ("s128x256", (("igemm8", 3),))]
                ab(f"dual {Ho} {C1}+{C2}->{K} B{B}", lambda: dual(B, Ho, C1, C2, K, 2), reps, v)
    if what in ("conv", "all"):
        for B in (256, 128):
            ab(f"r50 3x3 128 28 B{B}", lambda: conv(B, 28, 128, 128, 3), reps, variants)
            ab(f"r50 3x3 256 14 B{B}", lambda: conv(B, 14, 256, 256, 3), reps, variants)
            ab(f"r50 3x3 512 7 B{B}", lambda: conv(B, 7, 512, 512, 3), reps, variants)
            ab(f"r50 3x3s2 128 56 B{B}", lambda: conv(B, 56, 128, 128, 3, 2), reps, variants)
            ab(f"r50 3x3s2 256 28 B{B}", lambda: conv(B, 28, 256, 256, 3, 2), reps, variants)
            ab(f"r50 3x3s2 512 14 B{B}", lambda: conv(B, 14, 512, 512, 3, 2), reps, variants)
            ab(f"r50 1x1 512->128 28 B{B}", lambda: conv(B, 28, 512, 128, 1), reps, variants)
            ab(f"r50 1x1 1024->256 14 B{B}", lambda: conv(B, 14, 1024, 256, 1), reps, variants)
            ab(f"r50 1x1 256->1024 14 res B{B}", lambda: conv(B, 14, 256, 1024, 1, res=True), reps, variants)
            ab(f"r50 1x1 2048->512 7 B{B}", lambda: conv(B, 7, 2048, 512, 1), reps, variants)
            ab(f"r50 1x1 512->2048 7 res B{B}", lambda: conv(B, 7, 512, 2048, 1, res=True), reps, variants)


if __name__ == "__main__":
    main()
