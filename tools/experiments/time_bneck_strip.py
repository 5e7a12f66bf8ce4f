"""Time mv_bottleneck_strip_fwd (a whole 56x56 bottleneck in one launch) at B images; compare with the un-fused launches it
replaces (conv3x3c64 43 us + chain1x1 111 us at 128 images, profiles/r03/resnet50_per_launch.txt).
usage: time_bneck_strip.py [B]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L
B = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = torch.cuda.current_stream().cuda_stream
bf = lambda *sh: torch.randn(*sh, device="cuda").bfloat16()


def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


SKEWS = [int(a) for a in os.environ.get("SKEWS", "0").split(",")]
for dual, skew in [(d, k) for d in (0, 1) for k in SKEWS]:
    L.set_flag("strip_skew", skew)
    cin = 64 if dual else 256
    x = torch.relu(bf(B, 56, 56, cin))
    w1f, w2f, w3f = bf(2, cin // 16, 64, 8) / cin ** 0.5, bf(2, 9, 4, 64, 8) / 24.0, bf(8, 8 if dual else 4, 64, 8) / 8.0
    f = [torch.rand(n, device="cuda") + 0.5 for n in (64, 64, 64, 64, 256, 256)]
    y = torch.empty(B, 56, 56, 256, device="cuda", dtype=torch.bfloat16)

    def go():
        L.call("mv_bottleneck_strip_fwd", x.data_ptr(), w1f.data_ptr(), f[0].data_ptr(), f[1].data_ptr(), w2f.data_ptr(),
               f[2].data_ptr(), f[3].data_ptr(), w3f.data_ptr(), f[4].data_ptr(), f[5].data_ptr(), y.data_ptr(), B, 56, 56, cin, 64,
               256, dual, 1, s)
    us = t(go)
    M = B * 56 * 56
    flops = 2.0 * M * (cin * 64 * 1.25 + 576 * 64 + (128 if dual else 64) * 256)
    by = 2.0 * M * (cin * 1.25 + 256)
    if os.environ.get("EQV_LIB"):            # debug build: per-wave phase stamps
        import numpy as np
        prof = torch.zeros(B * 7 * 8 * 8, dtype=torch.int64, device="cuda")
        pp = prof.data_ptr(); lo = pp & 0xffffffff
        if lo >= 1 << 31: lo -= 1 << 32
        L.set_flag("prof_hi", pp >> 32); L.set_flag("prof_lo", lo); L.set_flag("bneck_prof", 1)
        go(); torch.cuda.synchronize()
        L.set_flag("bneck_prof", 0)
        a = prof.cpu().numpy().reshape(B * 7, 8, 8).astype(np.float64) / 100.0          # us
        t0 = a[:, :, 0].min(axis=1, keepdims=True)
        names = ["zero+dma issue", "phase A (conv1)", "wait barrier 1", "phase B (conv2)", "wait barrier 2 + W3 copy", "C block 0", "C block 1"]
        d = np.diff(a, axis=2)
        print("   per-wave phase durations (us), mean over workgroups:")
        for wv in (0, 3, 6, 7):
            print(f"   wave {wv}: " + "  ".join(f"{n} {d[:, wv, i].mean():5.2f}" for i, n in enumerate(names)) +
                  f"   | total {(a[:, wv, 7] - a[:, wv, 0]).mean():5.1f}")
        span = a[:, :, 7].max(axis=1) - a[:, :, 0].min(axis=1)
        print(f"   workgroup span mean {span.mean():.1f} us  p10 {np.percentile(span, 10):.1f}  p90 {np.percentile(span, 90):.1f};  launch span {(a[:, :, 7].max() - a[:, :, 0].min()):.1f} us")
    print(f"skew {skew / 100:.1f} us  {L.last_kernel()} B={B}: {us:.1f} us  {flops / us / 1e6:.0f} TFLOP/s  {by / us / 1e3:.0f} GB/s ({by / 1e6:.0f} MB: x with halo + y)")
