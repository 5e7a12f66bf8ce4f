"""Phase split (prologue / main loop / epilogue per tile) of igemm2 on the ResNet-50 convolution shapes."""
import sys
sys.path.insert(0, "/root/repo/tools"); sys.path.insert(0, "/root/repo")
import torch, numpy as np
from eqxvision_amd import _lib as L
def conv(N, H, C, K, R, stride=1, flags=()):
    pad = R // 2
    x = torch.randn(N, H, H, C, device="cuda").to(torch.bfloat16)
    w = (torch.randn(K, R, R, C, device="cuda") / (R * R * C) ** 0.5).to(torch.bfloat16)
    b = torch.randn(K, device="cuda"); sc = torch.rand(K, device="cuda") + 0.5
    Ho = (H + 2 * pad - R) // stride + 1
    y = torch.empty(N, Ho, Ho, K, device="cuda", dtype=torch.bfloat16)
    s = torch.cuda.current_stream().cuda_stream
    for f, v in flags: L.set_flag(f, v)
    def go():
        L.call("mv_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), b.data_ptr(), None, y.data_ptr(),
               N, H, H, C, K, R, R, stride, stride, pad, pad, 1, 1, 1, 1, 1, 1, s)
    for _ in range(3): go()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record(); go(); e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3
    prof = torch.zeros(1 << 16, dtype=torch.int64, device="cuda")
    p = prof.data_ptr(); lo = p & 0xffffffff
    if lo >= 1 << 31: lo -= 1 << 32
    L.set_flag("prof_lo", lo); L.set_flag("prof_hi", p >> 32)
    go(); torch.cuda.synchronize()
    L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
    k = L.last_kernel()
    for f, v in flags: L.set_flag(f, 0)
    a = prof.cpu().numpy().reshape(-1, 4); a = a[a[:, 0] > 0]
    if len(a) == 0: print(N, H, C, K, R, k, "not instrumented", us); return
    pro = (a[:, 1] - a[:, 0]) / 100.0; main = (a[:, 2] - a[:, 1]) / 100.0; epi = (a[:, 3] - a[:, 2]) / 100.0
    print(f"N{N} {H}x{H} C{C} K{K} {R}x{R} s{stride} {k}: {us:.1f} us blocks {len(a)}  per tile: pro {pro.mean():.2f} main {main.mean():.2f} epi {epi.mean():.2f} us")
conv(128, 14, 256, 256, 3)
conv(128, 28, 128, 128, 3)
conv(128, 7, 512, 512, 3)
conv(128, 56, 128, 128, 3, stride=2)
conv(128, 14, 1024, 256, 1)
conv(128, 7, 2048, 512, 1)
conv(128, 28, 512, 128, 1)
