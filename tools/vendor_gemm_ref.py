"""Yardstick only (never in the product path): what the vendor GEMM (torch.matmul -> hipBLASLt/rocBLAS) reaches on the
ViT / ResNet GEMM shapes on this box, next to mv_linear_fwd."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
from gemm_sweep_lib import run

def vendor(M, N, K):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    for _ in range(3): torch.matmul(a, w.t())
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    e0.record()
    for _ in range(10): torch.matmul(a, w.t())
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 10 * 1e3

for M, N, K in ((50432, 3072, 768), (50432, 768, 3072), (50432, 2304, 768), (50432, 768, 768), (50432, 4096, 4096),
                (200704, 512, 128), (200704, 128, 512), (50176, 1024, 256), (25088, 1536, 384), (8192, 8192, 8192)):
    v = vendor(M, N, K)
    us, k = run(M, N, K)
    us3, k3 = run(M, N, K, flags=(("igemm3", 2),))
    fl = 2.0 * M * N * K
    print(f"M{M} N{N} K{K}: vendor {v:8.1f} us {fl/v/1e6:7.1f} TF | mv {us:8.1f} us {fl/us/1e6:7.1f} TF ({k}) | "
          f"{us3:8.1f} us {fl/us3/1e6:7.1f} TF ({k3})", flush=True)
