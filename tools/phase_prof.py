"""Phase split of igemm2 tiles: prologue / main loop / epilogue, from per-block wall-clock stamps (100 MHz)."""
import sys, os
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import torch, numpy as np
from gemm_sweep_lib import L
def one(M, N, K, act=0, res=False, f32=False, conv=None, flags=()):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=odt)
    s = torch.cuda.current_stream().cuda_stream
    prof = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    for f, v in flags: L.set_flag(f, v)
    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), M, N, K, act, 1, 0 if f32 else 1, s)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    p = prof.data_ptr()
    L.set_flag("prof_lo", p & 0x7fffffff | 0); L.set_flag("prof_hi", p >> 32)
    # low half may have bit 31 set: pass as signed int
    lo = p & 0xffffffff
    if lo >= 1 << 31: lo -= 1 << 32
    L.set_flag("prof_lo", lo)
    go(); torch.cuda.synchronize()
    L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
    k = L.last_kernel()
    for f, v in flags: L.set_flag(f, 0)
    a = prof.cpu().numpy().reshape(-1, 4)
    a = a[a[:, 0] > 0]
    if len(a) == 0:
        print(f'M{M} N{N} K{K}: {k} not instrumented ({us:.1f} us)'); return
    t0 = a[:, 0].min()
    pro = (a[:, 1] - a[:, 0]) / 100.0; main = (a[:, 2] - a[:, 1]) / 100.0; epi = (a[:, 3] - a[:, 2]) / 100.0
    span = (a[:, 3].max() - t0) / 100.0
    # CU busy fraction: sum of block durations / (256 * span)
    busy = ((a[:, 3] - a[:, 0]) / 100.0).sum() / (256 * span)
    print(f"M{M} N{N} K{K} act{act} res{int(res)} f32{int(f32)} {k}: {us:.1f} us {2.0*M*N*K/us/1e6:.0f} TF  blocks {len(a)} "
          f"span {span:.1f}  per tile: pro {pro.mean():.2f} main {main.mean():.2f} epi {epi.mean():.2f} us  CU-busy {busy:.2f}")
B = 256
T = 197 * B
for fl in ((), (("igemm4", 2),), (("igemm4", 3),)):
    one(T, 2304, 768, flags=fl)
    one(T, 3072, 768, act=2, flags=fl)
    one(T, 768, 3072, res=True, f32=True, flags=fl)
    one(T, 768, 768, res=True, f32=True, flags=fl)
    one(8192, 8192, 8192, flags=fl)
