"""Time mv_ln_mlp_fwd (fused LayerNorm + MLP, Swin stage 0) against the three separate launches.  usage: time_ln_mlp.py [M]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L
M = int(sys.argv[1]) if len(sys.argv) > 1 else 64 * 56 * 56
C, H = 96, 384
s = torch.cuda.current_stream().cuda_stream
x = torch.randn(M, C, device="cuda")
w1 = (torch.randn(H, C, device="cuda") / C ** 0.5).bfloat16()
w2 = (torch.randn(C, H, device="cuda") / H ** 0.5).bfloat16()
b1, b2 = torch.randn(H, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
y = torch.empty_like(x)
nb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
hb = torch.empty(M, H, device="cuda", dtype=torch.bfloat16)

def fused():
    L.call("mv_ln_mlp_fwd", x.data_ptr(), w1.data_ptr(), b1.data_ptr(), w2.data_ptr(), b2.data_ptr(), y.data_ptr(), M, C, H, 1e-5, 0, s)

def unfused():
    L.call("mv_layernorm_fwd", x.data_ptr(), g.data_ptr(), be.data_ptr(), nb.data_ptr(), M, C, 0, 1e-5, 0, 1, s)
    L.call("mv_linear_fwd", nb.data_ptr(), w1.data_ptr(), None, b1.data_ptr(), None, hb.data_ptr(), M, H, C, 2, 1, 1, s)
    L.call("mv_linear_fwd", hb.data_ptr(), w2.data_ptr(), None, b2.data_ptr(), x.data_ptr(), y.data_ptr(), M, C, H, 0, 1, 0, s)

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for flags in ((), (("ln_mlp_waves", 8),), (("ln_mlp_waves", 16),)) + tuple(((k, v),) for k, v in [a.split("=") for a in sys.argv[2:]]):
    for k, v in flags: L.set_flag(k, int(v))
    us = t(fused)
    print(f"fused {dict(flags)}: {us:.1f} us  ({(M*C*8)/us/1e3:.0f} GB/s algorithmic, {4.0*M*C*H/us/1e6:.0f} TFLOP/s)")
    for k, v in flags: L.set_flag(k, 0)
print(f"unfused (LN + fc1 + fc2): {t(unfused):.1f} us")
