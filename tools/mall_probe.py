"""Does a ResNet layer-1 / layer-2 launch run faster when its tensors fit the 256 MB Infinity Cache?  The HBM-bound kernels of the
hot path (chain1x1, chain_stream, conv3x3c64, stream1x1) are timed at N = 8 .. 128 images on the SAME buffers in a loop: below
~100 MB per launch the whole working set stays in the memory-side cache, at 128 images (514 MB) it cannot.  Reported: us per launch,
us per image, algorithmic GB/s.  If the small batches run much faster per image, a depth-first schedule of the early stages
(sub-batches of 16-32 images through a whole stage) would turn HBM time into cache time."""
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


def chain(N, HW, C, K, N2):
    M = N * HW * HW
    x, r = bf(M, C), bf(M, K)
    w3, w1 = (torch.randn(K, C, device="cuda") / C ** 0.5).bfloat16(), (torch.randn(N2, K, device="cuda") / K ** 0.5).bfloat16()
    s3, h3, s1, h1 = (torch.rand(n, device="cuda") + 0.5 for n in (K, K, N2, N2))
    y, t1 = torch.empty(M, K, device="cuda", dtype=torch.bfloat16), torch.empty(M, N2, device="cuda", dtype=torch.bfloat16)
    if not L.load().mv_conv1x1_chain_supported(M, C, K, N2, 1): return None
    def go():
        L.call("mv_conv1x1_chain_fwd", x.data_ptr(), w3.data_ptr(), s3.data_ptr(), h3.data_ptr(), r.data_ptr(), y.data_ptr(),
               w1.data_ptr(), s1.data_ptr(), h1.data_ptr(), t1.data_ptr(), M, C, K, N2, 1, s)
    return t(go), 2.0 * M * (C + 2 * K + N2)


def conv(N, H, C, K, R, stride=1, res=False):
    pad = R // 2
    x = bf(N, H, H, C)
    w = (torch.randn(K, R, R, C, device="cuda") / (C * R * R) ** 0.5).bfloat16()
    sc = torch.rand(K, device="cuda") + 0.5; sh = torch.randn(K, device="cuda") * 0.1
    Ho = (H + 2 * pad - R) // stride + 1
    r = bf(N, Ho, Ho, K) if res else None
    y = torch.empty(N, Ho, Ho, K, device="cuda", dtype=torch.bfloat16)
    def go():
        L.call("mv_conv2d_nhwc_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sh.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), N, H, H, C, K, R, R, stride, stride, pad, pad, 1, 1, 1, 1, 1, 1, s)
    return t(go), 2.0 * (N * H * H * C + N * Ho * Ho * K * (2 if res else 1))


def lin(M, N_, K_, res=False, f32=False, act=0):
    x = bf(M, K_)
    w = (torch.randn(N_, K_, device="cuda") / K_ ** 0.5).bfloat16()
    b = torch.randn(N_, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N_, device="cuda").to(odt) if res else None
    y = torch.empty(M, N_, device="cuda", dtype=odt)
    def go():
        L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None if r is None else r.data_ptr(),
               y.data_ptr(), M, N_, K_, act, 1, 0 if f32 else 1, s)
    osz = 4 if f32 else 2
    return t(go), 2.0 * M * K_ + osz * M * N_ * (2 if res else 1)


def ln(M, C):
    x = torch.randn(M, C, device="cuda")
    g, b = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)
    def go():
        L.call("mv_layernorm_fwd", x.data_ptr(), g.data_ptr(), b.data_ptr(), y.data_ptr(), M, C, C, 1e-5, 0, 1, s)
    return t(go), 6.0 * M * C


def row(name, N, r):
    if r is None:
        print(f"{name:44s} N={N:4d}: unsupported"); return
    us, by = r
    print(f"{name:44s} N={N:4d}: {us:8.1f} us  {us / N:7.3f} us/img  {by / 1e6:7.1f} MB  {by / us / 1e3:6.0f} GB/s", flush=True)


if __name__ == "__main__":
    for N in (8, 16, 32, 64, 128):
        row("chain1x1 56x56 64->256(+res)->64", N, chain(N, 56, 64, 256, 64))
    for N in (8, 16, 32, 64, 128):
        row("conv3x3 56x56 64->64", N, conv(N, 56, 64, 64, 3))
    for N in (8, 16, 32, 64, 128):
        row("chain_stream 28x28 128->512(+res)->128", N, chain(N, 28, 128, 512, 128))
    for N in (8, 16, 32, 64, 128):
        row("conv3x3 28x28 128->128", N, conv(N, 28, 128, 128, 3))
    for N in (8, 16, 32, 64, 128):
        row("conv1x1 56x56 64->256 +res (stream1x1)", N, conv(N, 56, 64, 256, 1, res=True))
    for N in (16, 32, 64, 128):
        row("vit layernorm 768 (rows = 197 N)", N, ln(197 * N, 768))
    for N in (16, 32, 64, 128):
        row("vit proj 768->768 f32+res", N, lin(197 * N, 768, 768, res=True, f32=True))
    for N in (16, 32, 64, 128):
        row("vit qkv 768->2304 bf16", N, lin(197 * N, 2304, 768))
