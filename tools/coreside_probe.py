"""Can a streaming kernel run UNDER a GEMM?  igemm8's Linear variant (LIN) takes 8 waves x 224 VGPRs of a CU; the slim LayerNorm
needs 4 waves x 64 VGPRs and no LDS -- both fit on a CU at once.  Two streams: the ViT qkv / fc1 GEMM looped on one, LayerNorm
looped on the other; wall time of both together against the sum of each alone.  Variants: default kernels (235 / 90 VGPRs: they
cannot co-reside) vs LIN + slim."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L

M = 197 * 128
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
x = torch.randn(M, 768, device="cuda").to(torch.bfloat16)
bias = torch.randn(3072, device="cuda")
xf = torch.randn(M, 768, device="cuda")
g, b = torch.ones(768, device="cuda"), torch.zeros(768, device="cuda")
yl = torch.empty(M, 768, device="cuda", dtype=torch.bfloat16)


def gemm(N, stream):
    w = (torch.randn(N, 768, device="cuda") / 768 ** 0.5).to(torch.bfloat16)
    y = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    return lambda: L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, bias.data_ptr(), None, y.data_ptr(), M, N, 768, 0, 1, 1,
                          stream.cuda_stream)


def ln(stream):
    return lambda: L.call("mv_layernorm_fwd", xf.data_ptr(), g.data_ptr(), b.data_ptr(), yl.data_ptr(), M, 768, 768, 1e-5, 0, 1,
                          stream.cuda_stream)


def wall(fa, na, fb, nb):
    torch.cuda.synchronize()
    t = time.perf_counter()
    for i in range(max(na, nb)):
        if i < na: fa()
        if i < nb: fb()
    torch.cuda.synchronize()
    return (time.perf_counter() - t) * 1e6


for N in (2304, 3072):
    for name, flags in (("default (235 / 90 VGPRs)", (("no_i8_lin", 1), ("ln_slim", 0))), ("LIN + slim LN (224 / 64)", (("no_i8_lin", 0), ("ln_slim", 1))),
                        ("LIN + default LN", (("no_i8_lin", 0), ("ln_slim", 0)))):
        for f, v in flags: L.set_flag(f, v)
        ga, lb = gemm(N, sA), ln(sB)
        for _ in range(3): ga(); lb()
        kg = None
        ga(); kg = L.last_kernel(); lb(); kl = L.last_kernel()
        n = 40
        tg = wall(ga, n, lambda: None, 0) / n
        tl = wall(lambda: None, 0, lb, n) / n
        both = wall(ga, n, lb, n) / n
        both4 = wall(ga, n, lb, 4 * n) / n
        print(f"N={N} {name:28s} [{kg} | {kl}]: GEMM alone {tg:6.1f} us  LN alone {tl:6.1f} us  1 GEMM + 1 LN together {both:6.1f} us (sum {tg + tl:6.1f})"
              f"  1 GEMM + 4 LN {both4:6.1f} us (sum {tg + 4 * tl:6.1f})", flush=True)
        for f, v in flags: L.set_flag(f, 0)
