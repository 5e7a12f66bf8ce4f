"""The fp32 classifier head alone (mv_linear_fwd, fp32 in / out: skinny_linear_f32_mfma) for PMC passes: usage one_head.py [M K N]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L
M, K, N = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (128, 2048, 1000)
x = torch.randn(M, K, device="cuda"); w = torch.randn(N, K, device="cuda") / K ** 0.5; b = torch.randn(N, device="cuda")
y = torch.empty(M, N, device="cuda"); s = torch.cuda.current_stream().cuda_stream
go = lambda: L.call("mv_linear_fwd", x.data_ptr(), w.data_ptr(), None, b.data_ptr(), None, y.data_ptr(), M, N, K, 0, 0, 0, s)
for _ in range(3): go()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
torch.cuda.synchronize(); e0.record()
for _ in range(20): go()
e1.record(); torch.cuda.synchronize()
print(f"M{M} K{K} N{N} [{L.last_kernel()}]: {e0.elapsed_time(e1)/20*1e3:.1f} us")
