"""Times mv_fc_stream_fwd on the AlexNet / VGG classifier shapes.  usage: time_fc_stream.py [M]
(round 3, with experiment switches since removed: split counts 2 ... 32 -- the rule's choice is within 5 % of the best everywhere;
256-wide k-chunks with 16 weight fragments in flight -- no gain; the floor is ~12 us of fixed cost for the two launches)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L

M = int(sys.argv[1]) if len(sys.argv) > 1 else 128
s = torch.cuda.current_stream().cuda_stream
for (K, N) in ((9216, 4096), (4096, 4096), (4096, 1000), (25088, 4096)):
    NT = (N + 31) // 32
    x = torch.randn((M, K), device="cuda").to(torch.bfloat16)
    wf = torch.randn((NT, K // 16, 64, 8), device="cuda").to(torch.bfloat16)
    b = torch.zeros((N,), device="cuda")
    y = torch.empty((M, N), dtype=torch.bfloat16, device="cuda")
    out = []
    for S, ch in ((0, 0),):
        nbytes = int(L.load().mv_fc_stream_workspace(M, N, K))
        ws = torch.empty((nbytes // 4,), dtype=torch.float32, device="cuda")
        f = lambda: L.call("mv_fc_stream_fwd", x.data_ptr(), wf.data_ptr(), b.data_ptr(), y.data_ptr(), ws.data_ptr(), nbytes, M, N, K, 1, 1, 1, s)
        for _ in range(3): f()
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): f()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        out.append(f"S={S or 'rule'}{'/ch256' if ch else ''}: {us:6.1f} us ({2.0 * N * K / us / 1e6:5.2f} TB/s)")
    print(f"M={M} K={K} N={N}: " + "  ".join(out), flush=True)
