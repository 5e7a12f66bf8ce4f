"""Two-way split-K of the half-size igemm8 tiles (include/eqxvision_amd.h: mv_set_scratch) vs the un-split launch, alone on the
chip, at the ResNet-50 layer-4 and Swin-T stage-3 shapes of one graph lane (64 / 128 images).
usage: time_splitk.py            -> one line per shape: us un-split, us split, ratio, max |diff|, kernel"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L

s = torch.cuda.current_stream().cuda_stream
bf = lambda *sh: torch.randn(*sh, device="cuda").bfloat16()


def t(fn, n=30):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def conv(N, H, W, C, K, R, stride, pad, res=False, out_f32=False):
    x = torch.relu(bf(N, H, W, C)); w = bf(K, R, R, C) / (R * R * C) ** 0.5
    sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.1
    Ho = (H + 2 * pad - R) // stride + 1
    y = torch.empty(N, Ho, Ho, K, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    r = torch.randn_like(y) if res else None
    go = lambda: L.call("mv_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr() if res else None,
                        y.data_ptr(), N, H, W, C, K, R, R, stride, stride, pad, pad, 1, 1, 1, 1, 1, 0 if out_f32 else 1, s)
    return go, y, (N * Ho * Ho, K, R * R * C), 2.0 * N * Ho * Ho * K * R * R * C


def linear(M, K, N, res=False, out_f32=False, act=0):
    x = bf(M, K); w = bf(N, K) / K ** 0.5; b = torch.randn(N, device="cuda") * 0.1
    y = torch.empty(M, N, device="cuda", dtype=torch.float32 if out_f32 else torch.bfloat16)
    r = torch.randn_like(y) if res else None
    go = lambda: L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), r.data_ptr() if res else None, y.data_ptr(),
                        M, N, K, act, 1, 0 if out_f32 else 1, s)
    return go, y, (M, N, K), 2.0 * M * N * K


def lsplit(M, K, N):
    x = bf(M, K); w = bf(N, 2 * K) / K ** 0.5; b = torch.randn(N, device="cuda") * 0.1
    y = torch.empty(M, N, device="cuda", dtype=torch.float32)
    go = lambda: L.call("mv_linear_split_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), M, N, K, 0, 1, 0, s)
    return go, y, (M, N, 2 * K), 4.0 * M * N * K


SHAPES = []
for B in (64, 128):
    SHAPES += [(f"resnet L4 3x3 512->512 7x7 B{B}", conv(B, 7, 7, 512, 512, 3, 1, 1)),
               (f"resnet L4 1x1 2048->512 7x7 B{B}", conv(B, 7, 7, 2048, 512, 1, 1, 0)),
               (f"resnet L4 3x3 s2 512->512 14->7 B{B}", conv(B, 14, 14, 512, 512, 3, 2, 1)),
               (f"resnet L4 1x1 512->2048 7x7 +res B{B}", conv(B, 7, 7, 512, 2048, 1, 1, 0, res=True)),
               (f"resnet L3 1x1 1024->256 14x14 B{B}", conv(B, 14, 14, 1024, 256, 1, 1, 0))]
for B in (64,):
    M = 49 * B
    SHAPES += [(f"swin S3 proj 768->768 f32+res M{M}", linear(M, 768, 768, res=True, out_f32=True)),
               (f"swin S3 fc2 3072->768 f32+res M{M}", linear(M, 3072, 768, res=True, out_f32=True)),
               (f"swin S3 qkv 768->2304 M{M}", linear(M, 768, 2304)),
               (f"swin S3 fc1 768->3072 gelu M{M}", linear(M, 768, 3072, act=2)),
               (f"swin merge 1536->768 hi+lo M{M}", lsplit(M, 1536, 768)),
               (f"swin merge 768->384 hi+lo M{4 * M}", lsplit(4 * M, 768, 384))]

for name, (go, y, (M, N, kred), flops) in SHAPES:
    us0 = t(go); k0 = L.last_kernel(); y0 = y.clone()
    nb = int(L.load().mv_splitk_scratch_bytes(M, N, kred))
    if not nb:
        print(f"{name:44s} {us0:7.1f} us {flops / us0 / 1e6:6.0f} TF/s   (no split by the rule)   {k0}")
        continue
    ws = torch.zeros(nb, dtype=torch.uint8, device="cuda")

    def go2():
        L.call("mv_set_scratch", ws.data_ptr(), nb, s)
        go()
    us1 = t(go2); k1 = L.last_kernel()
    d = (y.float() - y0.float()).abs().max().item()
    print(f"{name:44s} {us0:7.1f} us {flops / us0 / 1e6:6.0f} TF/s -> {us1:7.1f} us {flops / us1 / 1e6:6.0f} TF/s  x{us0 / us1:.2f}  "
          f"max|diff| {d:.3g}  {k1}")
