"""Which vendor GEMM kernel (name encodes tile / MFMA / staging choices) does torch.matmul pick for the ViT shapes?
Run under: rocprofv3 --kernel-trace --stats --output-format csv -d DIR -o v -- python tools/vendor_kernel_names.py"""
import torch
for M, N, K in ((50432, 3072, 768), (50432, 768, 3072), (50432, 2304, 768), (50432, 768, 768), (8192, 8192, 8192)):
    a = torch.randn(M, K, device="cuda").to(torch.bfloat16)
    w = (torch.randn(N, K, device="cuda") / K ** 0.5).to(torch.bfloat16)
    for _ in range(5): torch.matmul(a, w.t())
    torch.cuda.synchronize()
