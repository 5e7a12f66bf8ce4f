"""Debug: where does the new igemm8 epilogue differ?  linear f32out + residual through mv_linear_fwd, new vs torch."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from eqxvision_amd import _lib as L
torch.manual_seed(0)
for (M, N, K, f32) in [(5880, 256, 512, True), (5880, 256, 512, False), (600, 256, 512, True)]:
    x = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    b = torch.randn(N, device="cuda")
    odt = torch.float32 if f32 else torch.bfloat16
    r = torch.randn(M, N, device="cuda").to(odt)
    y = torch.full((M + 64, N), 768.0, device="cuda", dtype=odt)
    s = torch.cuda.current_stream().cuda_stream
    L.set_flag("igemm8", 2)
    L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), r.data_ptr(), y.data_ptr(), M, N, K, 0, 1, 0 if f32 else 1, s)
    torch.cuda.synchronize()
    print(L.last_kernel())
    ref = x.float() @ w.float().t() + b + r.float()
    d = (y[:M].float() - ref).abs().cpu().numpy()
    bad = d > 0.05
    print(f"M{M} f32={f32}: max err {d.max():.3f}, bad {bad.sum()} of {bad.size}; guard rows intact: {(y[M:].float() == 768.0).all().item()}")
    if bad.any():
        rows = np.where(bad.any(1))[0]; cols = np.where(bad.any(0))[0]
        print(" bad rows: n", len(rows), rows[:40], "... mod 256:", sorted(set(rows % 256))[:64])
        print(" bad cols:", cols[:64])
        # is it y == ref - r (residual missing)?  or residual of another row?
        nores = x.float() @ w.float().t() + b
        dd = (y[:M].float() - nores).cpu().numpy()
        rr = r.float().cpu().numpy()
        i, j = np.argwhere(bad)[0]
        print(" first bad", i, j, "y-nores", dd[i, j], "r[i,j]", rr[i, j], "matches r at rows:", np.where(np.abs(rr[:, j] - dd[i, j]) < 1e-3)[0][:8])
L.set_flag("igemm8", 0)
