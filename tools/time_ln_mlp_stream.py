"""Time mv_ln_mlp_stream_fwd (LayerNorm + fc1 + GELU + fc2 + residual in one launch, weights streamed from L2) against the three
launches it replaces (LayerNorm, fc1 + GELU, fc2 + residual).  usage: time_ln_mlp_stream.py [M ...]  (default: Swin stage 2)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eqxvision_amd import _lib as L
from eqxvision_amd.ops import ln_mlp_fragments
C = int(os.environ.get('LMS_C', '384')); Hd = 4 * C
s = torch.cuda.current_stream().cuda_stream

def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for M in [int(a) for a in sys.argv[1:]] or ([64 * 196, 128 * 196] if C == 384 else [64 * 784]):
    x = torch.randn(M, C, device="cuda")
    w1 = (np.random.randn(Hd, C) / C ** 0.5).astype(np.float32); w2 = (np.random.randn(C, Hd) / Hd ** 0.5).astype(np.float32)
    w1f, w2f = ln_mlp_fragments(w1, w2)
    w1fd, w2fd = torch.from_numpy(w1f).cuda().bfloat16(), torch.from_numpy(w2f).cuda().bfloat16()
    w1d, w2d = torch.from_numpy(w1).cuda().bfloat16(), torch.from_numpy(w2).cuda().bfloat16()
    b1, b2, g, be = torch.randn(Hd, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1, torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
    y, y3 = torch.empty_like(x), torch.empty_like(x)
    nb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16); hb = torch.empty(M, Hd, device="cuda", dtype=torch.bfloat16)

    def fused():
        L.call("mv_ln_mlp_stream_fwd", x.data_ptr(), w1fd.data_ptr(), b1.data_ptr(), w2fd.data_ptr(), b2.data_ptr(), y.data_ptr(), M, C, Hd, 1e-5, 0, s)

    def unfused():
        L.call("mv_layernorm_fwd", x.data_ptr(), g.data_ptr(), be.data_ptr(), nb.data_ptr(), M, C, 0, 1e-5, 0, 1, s)
        L.call("mv_linear_fwd", nb.data_ptr(), w1d.data_ptr(), None, b1.data_ptr(), None, hb.data_ptr(), M, Hd, C, 2, 1, 1, s)
        L.call("mv_linear_fwd", hb.data_ptr(), w2d.data_ptr(), None, b2.data_ptr(), x.data_ptr(), y3.data_ptr(), M, C, Hd, 0, 1, 0, s)

    fl = 4.0 * M * C * Hd
    uf = t(fused); uu = t(unfused)
    fused(); unfused(); torch.cuda.synchronize()
    d = (y - y3).abs().max().item()
    print(f"M={M}: fused {uf:.1f} us ({fl/uf/1e6:.0f} TFLOP/s, C={C})   three launches {uu:.1f} us ({fl/uu/1e6:.0f} TFLOP/s)   max|diff| {d:.4f}")

    if os.environ.get("LMS_PROF"):
        nb_ = (M + 63) // 64
        prof = torch.zeros(nb_ * 8 * 18, dtype=torch.int64, device="cuda")
        pp = prof.data_ptr(); lo = pp & 0xffffffff
        if lo >= 1 << 31: lo -= 1 << 32
        L.set_flag("prof_hi", pp >> 32); L.set_flag("prof_lo", lo); L.set_flag("lms_prof", 1)
        fused(); torch.cuda.synchronize()
        L.set_flag("lms_prof", 0); L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
        a = prof.cpu().numpy().reshape(nb_, 8, 18)
        w = a[:, :, :9].astype(np.float64) / 100.0
        cy = a[:, :, 9:].astype(np.int64)
        t0 = w[:, :, 0].min()
        names = ["LayerNorm + barrier", "fc1(0)", "GELU(0) + barrier", "fc1(1)", "fc2(0) + GELU(1)", "barrier", "chunks 2..5 + fc2(5)", "epilogue"]
        print(f"  M={M}: kernel span {w[:, :, 8].max() - t0:.1f} us; WG start spread {w[:, :, 0].max() - t0:.1f} us; per-wave mean (min..max) us | cycles")
        for i, nm in enumerate(names):
            d = w[:, :, i + 1] - w[:, :, i]
            dc = (cy[:, :, i + 1] - cy[:, :, i]) & 0xffffffff
            print(f"    {nm:22s} {d.mean():7.2f} ({d.min():6.2f} .. {d.max():6.2f})   cycles {dc.mean():9.0f}  -> {dc.mean() / max(d.mean(), 1e-9) / 1e3:.2f} GHz")
