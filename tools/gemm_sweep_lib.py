"""GEMM sweep through mv_linear_fwd: time vs K for fixed M,N -> per-iteration slope and per-block intercept."""
import sys, os, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L

def run(M, N, K, act=0, res=False, f32=False, flags=()):
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt) if res else None
    y = torch.empty(M, N, device="cuda", dtype=odt)
    s = torch.cuda.current_stream().cuda_stream
    for f, v in flags: L.set_flag(f, v)
    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), M, N, K, act, 1, 0 if f32 else 1, s)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    n = 10
    for _ in range(n): go()
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / n * 1e3
    k = L.last_kernel()
    for f, v in flags: L.set_flag(f, 0)
    return us, k

