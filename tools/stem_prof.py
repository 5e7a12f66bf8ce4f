"""Time the fused ResNet entry (conv 7x7/2 + BN + ReLU + max-pool) on a half batch: usage stem_prof.py"""
import sys, os
sys.path.insert(0, "/root/repo")
import torch, numpy as np
from eqxvision_amd import _lib as L
N = 128
x = torch.rand(N, 3, 224, 224, device="cuda")
w = (torch.randn(64, 3, 7, 7, device="cuda") / 12).to(torch.bfloat16)
sc = torch.rand(64, device="cuda") + 0.5; sf = torch.randn(64, device="cuda") * 0.1
y = torch.empty(N, 56, 56, 64, dtype=torch.bfloat16, device="cuda")
s = torch.cuda.current_stream().cuda_stream
def go():
    L.call("mv_stem_conv_pool_fwd", x.data_ptr(), w.data_ptr(), sc.data_ptr(), sf.data_ptr(), y.data_ptr(),
           N, 3, 224, 224, 64, 7, 7, 2, 2, 3, 3, 3, 2, 1, 1, 0, 1, s)
for _ in range(3): go()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(10): go()
e1.record(); torch.cuda.synchronize()
print("us per launch (N=%d)" % N, e0.elapsed_time(e1) * 100)
