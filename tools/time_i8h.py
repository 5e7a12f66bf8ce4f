"""igemm8h (two workgroups per CU, 128 x 256 tile, 24 KB stages) against igemm8 on the hot dense shapes, in one process.
usage: time_i8h.py [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eqxvision_amd import _lib as L
S = lambda: torch.cuda.current_stream().cuda_stream


def timed(go, n=10):
    for _ in range(2): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): go()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def lin(M, N, K, act=0, res=False, f32=False, scale=False):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    sc = (torch.rand(N, device="cuda") + 0.5) if scale else None
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=odt)
    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None if sc is None else sc.data_ptr(), b.data_ptr(),
               None if r is None else r.data_ptr(), y.data_ptr(), M, N, K, act, 1, 0 if f32 else 1, S())
    return go, 2.0 * M * N * K, y


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
shapes = [("8192^3", (8192, 8192, 8192), {}), ("4096^3", (4096, 4096, 4096), {})]
for B in (128, 256):
    M = 197 * B
    shapes += [(f"vit qkv B{B}", (M, 2304, 768), {}), (f"vit proj f32+res B{B}", (M, 768, 768), dict(res=True, f32=True)),
               (f"vit fc1 gelu B{B}", (M, 3072, 768), dict(act=2)), (f"vit fc2 f32+res B{B}", (M, 768, 3072), dict(res=True, f32=True))]
shapes += [("r50 1x1 1024->256 14x14 B128", (128 * 196, 256, 1024), dict(act=1, scale=True)),
           ("r50 1x1 512->2048 7x7 B128 +res", (128 * 49, 2048, 512), dict(act=1, scale=True, res=True)),
           ("r50 1x1 2048->512 7x7 B128", (128 * 49, 512, 2048), dict(act=1, scale=True)),
           ("swin s2 fc1 384->1536 B64", (64 * 196, 1536, 384), dict(act=2)),
           ("swin s3 qkv 768->2304 B64", (64 * 49, 2304, 768), {}), ("swin s3 fc2 3072->768 f32res B64", (64 * 49, 768, 3072), dict(res=True, f32=True))]
for name, (M, N, K), kw in shapes:
    go, flops, y = lin(M, N, K, **kw)
    out, line = {}, f"{name:34s}"
    for v, flags in (("rule", ()), ("i8 256x256", (("igemm8", 2),)), ("i8h", (("igemm8", 2), ("i8h", 1)))):
        ts = []
        for f, val in flags: L.set_flag(f, val)
        for _ in range(reps): ts.append(timed(go))
        kern = L.last_kernel()
        out[v] = y.float().clone()
        for f, val in flags: L.set_flag(f, 0)
        us = float(np.median(ts))
        d = float((out[v] - out["rule"]).abs().max())
        line += f" | {v}: {kern[:28]:28s} {us:7.1f} us {flops / us / 1e6:5.0f} TF d={d:.2g}"
    print(line, flush=True)
