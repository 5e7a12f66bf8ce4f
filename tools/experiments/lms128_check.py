"""ln_mlp_stream C=384: the 128-row kernel (flag lms_tm128) against the 64-row kernel on the same inputs (they round at the same places)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eqxvision_amd import _lib as L
from eqxvision_amd.ops import ln_mlp_fragments
C, Hd = 384, 1536
s = torch.cuda.current_stream().cuda_stream
for M in (12544, 12544 - 37, 200, 128, 25088):
    x = torch.randn(M, C, device="cuda") * 2 + 0.3
    w1 = (np.random.randn(Hd, C) / C ** 0.5).astype(np.float32); w2 = (np.random.randn(C, Hd) / Hd ** 0.5).astype(np.float32)
    w1f, w2f = ln_mlp_fragments(w1, w2)
    w1d, w2d = torch.from_numpy(w1f).cuda().bfloat16(), torch.from_numpy(w2f).cuda().bfloat16()
    b1, b2 = torch.randn(Hd, device="cuda") * 0.1, torch.randn(C, device="cuda") * 0.1
    ys = []
    for flag in (0, 1):
        y = torch.full((M + 8, C), 7.0, device="cuda")
        L.set_flag("lms_tm128", flag)
        L.call("mv_ln_mlp_stream_fwd", x.data_ptr(), w1d.data_ptr(), b1.data_ptr(), w2d.data_ptr(), b2.data_ptr(), y.data_ptr(), M, C, Hd, 1e-5, 0, s)
        torch.cuda.synchronize()
        ys.append((y, L.last_kernel()))
    L.set_flag("lms_tm128", 0)
    d = (ys[0][0][:M] - ys[1][0][:M]).abs().max().item()
    print(f"M={M}: {ys[0][1]} vs {ys[1][1]}: max |diff| {d:.2e}; guard rows intact {bool((ys[1][0][M:] == 7.0).all())}; out scale {ys[0][0][:M].abs().max().item():.2f}")
