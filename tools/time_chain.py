"""Time mv_conv1x1_chain_fwd (bottleneck tail + next bottleneck head in one launch) against the un-fused pair of 1x1 layers.
usage: time_chain.py [C K N2 [M]]   (defaults: ResNet-50 layer2 at 256 images: 128 512 128 256*28*28)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L
C, K, N2 = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 512, 128)
M = int(sys.argv[4]) if len(sys.argv) > 4 else 256 * 28 * 28
s = torch.cuda.current_stream().cuda_stream
bf = lambda *sh: torch.randn(*sh, device="cuda").bfloat16()
x, r = bf(M, C), bf(M, K)
w3, w1 = (torch.randn(K, C, device="cuda") / C ** 0.5).bfloat16(), (torch.randn(N2, K, device="cuda") / K ** 0.5).bfloat16()
s3, h3, s1, h1 = (torch.rand(n, device="cuda") + 0.5 for n in (K, K, N2, N2))
y, t1 = torch.empty(M, K, device="cuda", dtype=torch.bfloat16), torch.empty(M, N2, device="cuda", dtype=torch.bfloat16)

def fused():
    L.call("mv_conv1x1_chain_fwd", x.data_ptr(), w3.data_ptr(), s3.data_ptr(), h3.data_ptr(), r.data_ptr(), y.data_ptr(),
           w1.data_ptr(), s1.data_ptr(), h1.data_ptr(), t1.data_ptr(), M, C, K, N2, 1, s)

def unfused():
    L.call("mv_linear_fwd", x.data_ptr(), w3.data_ptr(), s3.data_ptr(), h3.data_ptr(), r.data_ptr(), y.data_ptr(), M, K, C, 1, 1, 1, s)
    L.call("mv_linear_fwd", y.data_ptr(), w1.data_ptr(), s1.data_ptr(), h1.data_ptr(), None, t1.data_ptr(), M, N2, K, 1, 1, 1, s)

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

if not L.load().mv_conv1x1_chain_supported(M, C, K, N2, 1):
    sys.exit("mv_conv1x1_chain_supported says no")
us = t(fused)
by = 2.0 * M * (C + 2 * K + N2)
print(f"fused   [{L.last_kernel()}]: {us:.1f} us  ({by/us/1e3:.0f} GB/s algorithmic, {2.0*M*K*(C+N2)/us/1e6:.0f} TFLOP/s)")
us = t(unfused)
print(f"unfused (two launches): {us:.1f} us  ({2.0*M*(C+3*K+N2)/us/1e3:.0f} GB/s of their own traffic)")
for extra in sys.argv[5:]:                      # e.g. chain_stream_w4=1: the same launch under a library switch
    k, v = extra.split("=")
    L.set_flag(k, int(v))
    us = t(fused)
    print(f"fused {extra} [{L.last_kernel()}]: {us:.1f} us  ({by/us/1e3:.0f} GB/s algorithmic)")
    L.set_flag(k, 0)
