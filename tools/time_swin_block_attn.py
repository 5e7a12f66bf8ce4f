"""Time mv_swin_block_attn_fwd (a Swin block's attention half in one launch, one workgroup per window) against the four launches it
replaces (LayerNorm, qkv Linear, window attention, proj Linear + residual).  usage: time_swin_block_attn.py [B ...]  (stage 2)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eqxvision_amd import _lib as L
from eqxvision_amd.ops import swin_block_attn_fragments
C = int(os.environ.get("SBA_C", "384")); heads = C // 32; Hf = {384: 14, 192: 28, 96: 56}[C]; ws, shift = 7, 3
s = torch.cuda.current_stream().cuda_stream
for _f in [a for a in os.environ.get("FLAGS", "").split(",") if a]:
    L.set_flag(_f.split("=")[0], int(_f.split("=")[1]) if "=" in _f else 1)

def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

rng = np.random.default_rng(0)
wq = (rng.standard_normal((3 * C, C)) / C ** 0.5).astype(np.float32); bq = (0.1 * rng.standard_normal(3 * C)).astype(np.float32)
wp = (rng.standard_normal((C, C)) / C ** 0.5).astype(np.float32); bp = (0.1 * rng.standard_normal(C)).astype(np.float32)
bias = (0.5 * rng.standard_normal((heads, 49, 49))).astype(np.float32)
wf, bqf, wpf, b64 = swin_block_attn_fragments(wq, bq, wp, bias)
cu = lambda a, dt=torch.float32: torch.from_numpy(np.ascontiguousarray(a)).cuda().to(dt)
wfd, bqfd, wpfd, bpd, b64d = cu(wf, torch.bfloat16), cu(bqf), cu(wpf, torch.bfloat16), cu(bp), cu(b64)
wqd, bqd, wpd, biasd = cu(wq, torch.bfloat16), cu(bq), cu(wp, torch.bfloat16), cu(bias)
g, be = torch.ones(C, device="cuda"), torch.zeros(C, device="cuda")
for B in [int(a) for a in sys.argv[1:]] or [64, 128]:
    M = B * Hf * Hf
    x = torch.randn(B, Hf, Hf, C, device="cuda")
    y, y4 = torch.empty_like(x), torch.empty_like(x)
    nb = torch.empty(M, C, device="cuda", dtype=torch.bfloat16); qd = torch.empty(M, 3 * C, device="cuda", dtype=torch.bfloat16)
    ad = torch.empty(M, C, device="cuda", dtype=torch.bfloat16)

    def fused():
        L.call("mv_swin_block_attn_fwd", x.data_ptr(), wfd.data_ptr(), bqfd.data_ptr(), wpfd.data_ptr(), bpd.data_ptr(), b64d.data_ptr(),
               y.data_ptr(), B, Hf, Hf, C, heads, ws, ws, shift, shift, 1e-5, 0, s)

    def unfused():
        L.call("mv_layernorm_fwd", x.data_ptr(), g.data_ptr(), be.data_ptr(), nb.data_ptr(), M, C, 0, 1e-5, 0, 1, s)
        L.call("mv_linear_fwd", nb.data_ptr(), wqd.data_ptr(), None, bqd.data_ptr(), None, qd.data_ptr(), M, 3 * C, C, 0, 1, 1, s)
        L.call("mv_swin_window_attn_fwd", qd.data_ptr(), biasd.data_ptr(), ad.data_ptr(), B, Hf, Hf, C, heads, ws, ws, shift, shift, 1, s)
        L.call("mv_linear_fwd", ad.data_ptr(), wpd.data_ptr(), None, bpd.data_ptr(), x.data_ptr(), y4.data_ptr(), M, C, C, 0, 1, 0, s)

    uf, uu = t(fused), t(unfused)
    fused(); unfused(); torch.cuda.synchronize()
    print(f"C={C} B={B} ({B * (Hf // 7) ** 2} windows): fused {uf:.1f} us   four launches {uu:.1f} us   max|diff| {(y - y4).abs().max().item():.4f}")

    if os.environ.get("SBA_PROF"):
        nwin = {384: 1, 192: 2, 96: 2}[C]; nwv = 4 if C == 96 else 8
        new96 = C == 96 and not L.get_flag("swin_c96_shared")            # swin_win96_kernel: one window, two waves, its own stamp list
        if new96:
            nwin, nwv = 1, 2
        nwg = B * (Hf // 7) ** 2 // nwin
        prof = torch.zeros(nwg * nwv * 9, dtype=torch.int64, device="cuda")
        pp = prof.data_ptr(); lo = pp & 0xffffffff
        if lo >= 1 << 31: lo -= 1 << 32
        L.set_flag("prof_hi", pp >> 32); L.set_flag("prof_lo", lo); L.set_flag("sba_prof", 1)
        fused(); torch.cuda.synchronize()
        L.set_flag("sba_prof", 0); L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
        a = prof.cpu().numpy().reshape(nwg, nwv, 9).astype(np.float64) / 100.0
        t0 = a[:, :, 0].min()
        if new96:
            a = a[:, :, :8]
            t0 = a[:, :, 0].min()
            names = ["gather + LayerNorm + residual seed", "q | k | v of 3 heads + k / v stores", "barrier", "head 0: attention + proj slice", "head 1", "head 2",
                     "epilogue stores (drained)"]
            print(f"  kernel span {a[:, :, 7].max() - t0:.1f} us; {nwg} workgroups; start times: median {np.median(a[:, 0, 0]) - t0:.1f} us, last {a[:, 0, 0].max() - t0:.1f} us; per-wave mean (min..max) us")
            for i, nm in enumerate(names):
                d = a[:, :, i + 1] - a[:, :, i]
                print(f"    {nm:36s} {d.mean():7.2f} ({d.min():6.2f} .. {d.max():6.2f})")
            d = a[:, :, 7] - a[:, :, 0]
            print(f"    {'workgroup total':36s} {d.mean():7.2f} ({d.min():6.2f} .. {d.max():6.2f})")
            continue
        names = ["gather + LayerNorm + barrier", "qkv GEMM (group 0)", "q/k/v stores + barrier", "attention (group 0)", "barrier", "groups 1, 2", "proj + staging + barrier", "epilogue"]
        print(f"  kernel span {a[:, :, 8].max() - t0:.1f} us; {nwg} workgroups; start times: median {np.median(a[:, 0, 0]) - t0:.1f} us, last {a[:, 0, 0].max() - t0:.1f} us; per-wave mean (min..max) us")
        for i, nm in enumerate(names):
            d = a[:, :, i + 1] - a[:, :, i]
            print(f"    {nm:30s} {d.mean():7.2f} ({d.min():6.2f} .. {d.max():6.2f})")
        d = a[:, :, 8] - a[:, :, 0]
        print(f"    {'workgroup total':30s} {d.mean():7.2f} ({d.min():6.2f} .. {d.max():6.2f})")
