"""Premise check (round 5): the two lanes' Swin stage-2 kernels (64 images each) on two streams at once vs one after the other.
ln_mlp_stream C=384 launches 196 workgroups per lane, swin_block_attn C=384 256: can the lanes overlap there at all?"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eqxvision_amd import _lib as L
from eqxvision_amd.ops import ln_mlp_fragments
C, Hd = 384, 1536
def mk(M):
    x = torch.randn(M, C, device="cuda")
    w1 = (np.random.randn(Hd, C) / C ** 0.5).astype(np.float32); w2 = (np.random.randn(C, Hd) / Hd ** 0.5).astype(np.float32)
    w1f, w2f = ln_mlp_fragments(w1, w2)
    return dict(x=x, y=torch.empty_like(x), w1=torch.from_numpy(w1f).cuda().bfloat16(), w2=torch.from_numpy(w2f).cuda().bfloat16(),
                b1=torch.randn(Hd, device="cuda") * 0.1, b2=torch.randn(C, device="cuda") * 0.1, M=M)
def go(d, s):
    L.call("mv_ln_mlp_stream_fwd", d["x"].data_ptr(), d["w1"].data_ptr(), d["b1"].data_ptr(), d["w2"].data_ptr(), d["b2"].data_ptr(),
           d["y"].data_ptr(), d["M"], C, Hd, 1e-5, 0, s)
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
def timed(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
for M in (64 * 196, 32 * 196):
    a, b = mk(M), mk(M)
    # parity of whichever kernel the flag selects against torch (fp32 math on the bf16-rounded operands is not reproduced: loose bound)
    go(a, torch.cuda.current_stream().cuda_stream); torch.cuda.synchronize()
    print("   kernel:", L.last_kernel(), " y finite:", bool(torch.isfinite(a["y"]).all()))
    cur = torch.cuda.current_stream()
    one = timed(lambda: go(a, cur.cuda_stream))
    seq = timed(lambda: (go(a, cur.cuda_stream), go(b, cur.cuda_stream)))
    def both():
        s0.wait_stream(cur); s1.wait_stream(cur)
        go(a, s0.cuda_stream); go(b, s1.cuda_stream)
        cur.wait_stream(s0); cur.wait_stream(s1)
    par = timed(both)
    # the same three patterns as hipGraphs (no host time between launches)
    res = []
    for fn in (lambda: go(a, torch.cuda.current_stream().cuda_stream),
               lambda: (go(a, torch.cuda.current_stream().cuda_stream), go(b, torch.cuda.current_stream().cuda_stream)),
               None):
        g = torch.cuda.CUDAGraph()
        cs = torch.cuda.Stream()
        with torch.cuda.stream(cs):
            with torch.cuda.graph(g, stream=cs):
                if fn is not None:
                    fn()
                else:
                    c2 = torch.cuda.current_stream()
                    s0.wait_stream(c2); s1.wait_stream(c2)
                    go(a, s0.cuda_stream); go(b, s1.cuda_stream)
                    c2.wait_stream(s0); c2.wait_stream(s1)
        res.append(timed(g.replay))
    print(f"ln_mlp_stream C=384 M={M} ({(M + 63) // 64} workgroups): eager one {one:.1f} / two in a row {seq:.1f} / two streams {par:.1f} us;"
          f"  as graphs: one {res[0]:.1f} / two in a row {res[1]:.1f} / two branches {res[2]:.1f} us")
