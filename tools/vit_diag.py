"""Round-4 diagnosis of the ViT GEMM tiles (debug build: EQV_PROF=1 python -m eqxvision_amd.build; run with
EQV_LIB=eqxvision_amd/csrc/libeqxvision_amd_prof.so).  Question: is a tile's epilogue slow because all CUs flush at once
(HBM-bound burst) or by itself (latency / issue)?  Per-tile prologue / main / epilogue stamps of igemm8's 256 x 256 kernel
  * on a full grid (M = 25216: 99 row tiles) and on a grid of 8 row tiles (24-96 workgroups: no HBM contention),
  * with the first round of workgroups de-phased by quarters (i8_skew),
  * with the epilogue's stores / residual loads ablated (i8_ablate 4 / 8 / 12; results wrong)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
from eqxvision_amd import _lib as L


def one(tag, M, N, K, act=0, res=False, f32=False, flags=()):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=odt)
    s = torch.cuda.current_stream().cuda_stream
    prof = torch.zeros(1 << 17, dtype=torch.int64, device="cuda")
    flags = (("igemm8", 2),) + tuple(flags)
    for f, v in flags: L.set_flag(f, v)
    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), M, N, K, act, 1, 0 if f32 else 1, s)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(5): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 5
    p = prof.data_ptr()
    lo = p & 0xffffffff
    if lo >= 1 << 31: lo -= 1 << 32
    L.set_flag("prof_hi", p >> 32); L.set_flag("prof_lo", lo)
    go(); torch.cuda.synchronize()
    L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
    for f, v in flags: L.set_flag(f, 0)
    nb = ((M + 255) // 256) * ((N + 255) // 256)
    raw = prof.cpu().numpy()
    a = raw[:4 * nb].reshape(-1, 4)
    a = a[a[:, 0] > 0]
    if len(a) == 0:
        print(f"{tag}: not instrumented ({us:.1f} us)"); return
    t0 = a[:, 0].min()
    pro = (a[:, 1] - a[:, 0]) / 100.0; main = (a[:, 2] - a[:, 1]) / 100.0; epi = (a[:, 3] - a[:, 2]) / 100.0
    span = (a[:, 3].max() - t0) / 100.0
    # epilogue start times relative to the launch: how synchronised are the flushes?
    es = np.sort((a[:, 2] - t0) / 100.0)
    first = es[: min(256, len(es))]
    print(f"{tag:34s} M{M} N{N} K{K}: {us:7.1f} us {2.0*M*N*K/us/1e6:5.0f} TF  tiles {len(a):4d} span {span:6.1f} | pro {pro.mean():5.2f} "
          f"main {main.mean():6.2f} epi {epi.mean():5.2f} (p10 {np.percentile(epi,10):5.2f} p90 {np.percentile(epi,90):5.2f}) us"
          f" | first-round epilogue starts: p5 {np.percentile(first,5):5.1f} p50 {np.percentile(first,50):5.1f} p95 {np.percentile(first,95):5.1f}", flush=True)


if __name__ == "__main__":
    T = 197 * 128
    shapes = [("qkv bf16", 2304, 768, dict()),
              ("proj f32+res", 768, 768, dict(res=True, f32=True)),
              ("proj bf16 no-res (delta out)", 768, 768, dict()),
              ("proj f32 no-res", 768, 768, dict(f32=True)),
              ("fc1 gelu bf16", 3072, 768, dict(act=2)),
              ("fc1 no-act bf16", 3072, 768, dict()),
              ("fc2 f32+res", 768, 3072, dict(res=True, f32=True)),
              ("fc2 bf16 no-res (delta out)", 768, 3072, dict())]
    for name, N, K, kw in shapes:
        print(f"--- {name}")
        one("full grid", T, N, K, **kw)
        one("full grid, B=256", 2 * T, N, K, **kw)
        one("8 row tiles", 2048, N, K, **kw)
        for sk in (230, 460, 700):
            one(f"full grid skew {sk/100:.1f} us x q", T, N, K, flags=(("i8_skew", sk),), **kw)
        one("full grid, no stores", T, N, K, flags=(("i8_ablate", 4),), **kw)
        if kw.get("res"):
            one("full grid, no residual", T, N, K, flags=(("i8_ablate", 8),), **kw)
            one("full grid, no stores no residual", T, N, K, flags=(("i8_ablate", 12),), **kw)
