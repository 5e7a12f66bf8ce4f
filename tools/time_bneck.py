"""Time mv_bottleneck_tail_fwd (conv2 3x3 + conv3 1x1 + identity of a ResNet identity bottleneck in one launch, one workgroup per
image) against the un-fused pair of launches (mv_conv2d_nhwc_fwd x 2).  usage: time_bneck.py [B ...]  (default 128 256)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L
HW, WID, COUT = 14, 256, 1024
s = torch.cuda.current_stream().cuda_stream
bf = lambda *sh: torch.randn(*sh, device="cuda").bfloat16()

def t(fn, n=30):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for B in [int(a) for a in sys.argv[1:]] or [128, 256]:
    t1, r = bf(B, HW, HW, WID).relu(), bf(B, HW, HW, COUT)
    w2 = (torch.randn(WID, 3, 3, WID, device="cuda") / (9 * WID) ** 0.5).bfloat16()          # KRSC
    w3 = (torch.randn(COUT, WID, device="cuda") / WID ** 0.5).bfloat16()
    w2f = w2.reshape(WID // 32, 32, 9, WID // 16, 2, 8).permute(0, 2, 3, 4, 1, 5).contiguous()
    w3f = w3.reshape(COUT // 256, 8, 32, WID // 16, 2, 8).permute(0, 1, 3, 4, 2, 5).contiguous()
    s2, h2, s3, h3 = (torch.rand(n, device="cuda") + 0.5 for n in (WID, WID, COUT, COUT))
    y, y2 = torch.empty(B, HW, HW, COUT, device="cuda", dtype=torch.bfloat16), torch.empty(B, HW, HW, COUT, device="cuda", dtype=torch.bfloat16)
    t2 = torch.empty(B, HW, HW, WID, device="cuda", dtype=torch.bfloat16)

    def fused():
        L.call("mv_bottleneck_tail_fwd", t1.data_ptr(), w2f.data_ptr(), s2.data_ptr(), h2.data_ptr(), w3f.data_ptr(), s3.data_ptr(),
               h3.data_ptr(), r.data_ptr(), y.data_ptr(), B, HW, HW, WID, COUT, 1, s)

    def unfused():
        L.call("mv_conv2d_nhwc_fwd", t1.data_ptr(), w2.data_ptr(), s2.data_ptr(), h2.data_ptr(), None, t2.data_ptr(),
               B, HW, HW, WID, WID, 3, 3, 1, 1, 1, 1, 1, 1, 1, 1, 1, 1, s)
        L.call("mv_conv2d_nhwc_fwd", t2.data_ptr(), w3.data_ptr(), s3.data_ptr(), h3.data_ptr(), r.data_ptr(), y2.data_ptr(),
               B, HW, HW, WID, COUT, 1, 1, 1, 1, 0, 0, 1, 1, 1, 1, 1, 1, s)

    fl = 2.0 * B * HW * HW * WID * (9 * WID + COUT)
    by = 2.0 * B * HW * HW * (WID + 2 * COUT)
    uf = t(fused)
    fused(); unfused(); torch.cuda.synchronize()
    d = (y.float() - y2.float()).abs().max().item()
    print(f"B={B} fused   [{L.last_kernel() if False else 'bneck_tail'}]: {uf:.1f} us  {fl/uf/1e6:.0f} TFLOP/s  {by/uf/1e3:.0f} GB/s algorithmic   max|fused-unfused|={d:.4f}")
    uu = t(unfused)
    print(f"B={B} unfused (two launches): {uu:.1f} us  {fl/uu/1e6:.0f} TFLOP/s")

    if os.environ.get("BNECK_PROF"):
        import numpy as np
        prof = torch.zeros(B * 8 * 12, dtype=torch.int64, device="cuda")
        pp = prof.data_ptr(); lo = pp & 0xffffffff
        if lo >= 1 << 31: lo -= 1 << 32
        L.set_flag("prof_hi", pp >> 32); L.set_flag("prof_lo", lo); L.set_flag("bneck_prof", 1)
        fused(); torch.cuda.synchronize()
        L.set_flag("bneck_prof", 0); L.set_flag("prof_lo", 0); L.set_flag("prof_hi", 0)
        a = prof.cpu().numpy().reshape(B, 8, 12)
        w = a[:, :, :6].astype(np.float64) / 100.0          # us (100 MHz wall clock)
        c = a[:, :, 6:].astype(np.int64)
        t0 = w[:, :, 0].min()
        names = ["phase0 (t1->LDS)", "conv2 main", "conv2 epi", "conv3 chunk0", "conv3 chunks1-3"]
        print(f"  B={B}: kernel span {w[:, :, 5].max() - t0:.1f} us; WG start spread {w[:, :, 0].max() - t0:.1f} us; per-wave mean (min..max) per phase, us | shader cycles:")
        for i, nm in enumerate(names):
            d = w[:, :, i + 1] - w[:, :, i]
            dc = (c[:, :, i + 1] - c[:, :, i]) & 0xffffffff
            print(f"    {nm:18s} {d.mean():7.2f} ({d.min():6.2f} .. {d.max():6.2f})   cycles {dc.mean():9.0f}  -> {dc.mean() / max(d.mean(), 1e-9) / 1e3:.2f} GHz")

if os.environ.get("BNECK_2STREAM"):
    # two half-batches on two streams, free-running: in phase (both start together) and offset by roughly half a launch
    B = 128
    def mk():
        t1, r = bf(B, HW, HW, WID).relu(), bf(B, HW, HW, COUT)
        y = torch.empty(B, HW, HW, COUT, device="cuda", dtype=torch.bfloat16)
        return t1, r, y
    bufs = [mk(), mk()]
    streams = [torch.cuda.Stream(), torch.cuda.Stream()]
    def go(i, n):
        t1, r, y = bufs[i]
        st = streams[i].cuda_stream
        for _ in range(n):
            L.call("mv_bottleneck_tail_fwd", t1.data_ptr(), w2f.data_ptr(), s2.data_ptr(), h2.data_ptr(), w3f.data_ptr(), s3.data_ptr(),
                   h3.data_ptr(), r.data_ptr(), y.data_ptr(), B, HW, HW, WID, COUT, 1, st)
    for off in (0, 32, 64):
        go(0, 3); go(1, 3); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        n = 40
        e0.record(streams[0]); streams[1].wait_event(e0)
        if off:
            t1, r, y = bufs[1]
            L.call("mv_bottleneck_tail_fwd", t1.data_ptr(), w2f.data_ptr(), s2.data_ptr(), h2.data_ptr(), w3f.data_ptr(), s3.data_ptr(),
                   h3.data_ptr(), r.data_ptr(), y.data_ptr(), off, HW, HW, WID, COUT, 1, streams[1].cuda_stream)
        go(0, n); go(1, n)
        d = torch.cuda.Event(); d.record(streams[1]); streams[0].wait_event(d); e1.record(streams[0]); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"two streams x B=128, offset launch of {off} images: {us:.1f} us per pair (256 images)  {fl*0+2.0*256*HW*HW*WID*(9*WID+COUT)/us/1e6:.0f} TFLOP/s")
