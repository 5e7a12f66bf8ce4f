"""Alone timing of the LayerNorm-fold Linear pair against the plain Linears (same shapes): usage time_ln_fold.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
M, D, H = B * 197, 768, 3072
L = _lib
st = torch.cuda.current_stream().cuda_stream
g = torch.Generator(device="cuda").manual_seed(0)
def rnd(*s, dt=torch.bfloat16, sc=1.0): return (torch.randn(*s, device="cuda", generator=g) * sc).to(dt)
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
xb, hb = rnd(M, D), rnd(M, H)
res = rnd(M, D, dt=torch.float32, sc=2.0)
y, yb, ylo = torch.empty(M, D, device="cuda"), torch.empty(M, D, device="cuda", dtype=torch.bfloat16), torch.empty(M, D, device="cuda", dtype=torch.bfloat16)
rhi, rlo = res.to(torch.bfloat16), (res - res.to(torch.bfloat16).float()).to(torch.bfloat16)
stats = torch.empty((D + 255) // 256, M, 2, device="cuda")
for name, x, K in (("proj", xb, D), ("fc2", hb, H)):
    w, b = rnd(D, K, sc=K ** -0.5), rnd(D, dt=torch.float32, sc=0.1)
    plain = timeit(lambda: L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), res.data_ptr(), y.data_ptr(), M, D, K, 0, 1, 0, st))
    line = f"{name:5s} f32out+res plain {plain:7.1f} us ({L.last_kernel()})  lnout:"
    t = timeit(lambda: L.call("mv_linear_lnout_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), res.data_ptr(), None, yb.data_ptr(), ylo.data_ptr(), stats.data_ptr(), M, D, K, 1, st))
    line += f"  fp32 rows -> planes {t:.1f}"
    t = timeit(lambda: L.call("mv_linear_lnout_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), rhi.data_ptr(), rlo.data_ptr(), yb.data_ptr(), ylo.data_ptr(), stats.data_ptr(), M, D, K, 1, st))
    line += f"  planes -> planes {t:.1f}"
    t = timeit(lambda: L.call("mv_linear_lnout_fwd", x.data_ptr(), w.data_ptr(), b.data_ptr(), rhi.data_ptr(), rlo.data_ptr(), y.data_ptr(), None, None, M, D, K, 1, st))
    line += f"  planes -> fp32 rows {t:.1f}"
    print(line, flush=True)
L.call("mv_linear_lnout_fwd", xb.data_ptr(), rnd(D, D, sc=D ** -0.5).data_ptr(), None, res.data_ptr(), None, yb.data_ptr(), ylo.data_ptr(), stats.data_ptr(), M, D, D, 1, st)
xn = rnd(M, D)
for name, N, act, tok in (("qkv", 3 * D, 0, 197), ("fc1", H, 2, 0)):
    w, b, cs = rnd(N, D, sc=D ** -0.5), rnd(N, dt=torch.float32, sc=0.1), rnd(N, dt=torch.float32, sc=0.1)
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    if tok:
        plain = timeit(lambda: L.call("mv_linear_heads_fwd", xn.data_ptr(), w.data_ptr(), None, b.data_ptr(), out.data_ptr(), M, N, D, tok, 64, 1, st))
    else:
        plain = timeit(lambda: L.call("mv_linear_fwd", xn.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, out.data_ptr(), M, N, D, act, 1, 1, st))
    k = L.last_kernel()
    line = f"{name:5s} plain {plain:7.1f} us ({k})  lnin:"
    for src, stag in ((yb, "stream rows"), (xn, "N(0,1) rows")):
        t = timeit(lambda: L.call("mv_linear_lnin_fwd", src.data_ptr(), stats.data_ptr(), w.data_ptr(), cs.data_ptr(), b.data_ptr(), out.data_ptr(), M, N, D, 1e-6, act, tok, 64 if tok else 0, 1, st))
        line += f"  {stag} {t:.1f}"
    print(line, flush=True)
xf = rnd(M, D, dt=torch.float32, sc=2.0)
gam, bet = torch.ones(D, device="cuda"), torch.zeros(D, device="cuda")
t = timeit(lambda: L.call("mv_layernorm_fwd", xf.data_ptr(), gam.data_ptr(), bet.data_ptr(), yb.data_ptr(), M, D, 0, 1e-6, 0, 1, st))
print(f"layernorm fp32 -> bf16 {t:.1f} us ({L.last_kernel()})")
