"""us per launch of the pieces a live-dropout ViT-B block adds (B images): attention with dropout, per-sample and per-token Dropout."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import eqxvision_amd as eqv
from eqxvision_amd import ops, _lib
from eqxvision_amd._act import Act
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
eqv.set_compute_dtype("bf16")
N, H, dh = 197, 12, 64
keys = eqv.random.split(eqv.random.PRNGKey(0), B)


def clock(f, n=5):
    for _ in range(2): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): f()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e6


qkv = Act(torch.randn((B, N, 3 * H * dh), device="cuda").bfloat16(), "seq", True)
print(f"mha              {clock(lambda: ops.mha(qkv, H, 0.125, False)):9.0f} us   [{_lib.last_kernel()}]")
print(f"mha + dropout    {clock(lambda: ops.mha(qkv, H, 0.125, False, drop=(0.1, keys))):9.0f} us   [{_lib.last_kernel()}]")
for C in (768, 3072):
    x = Act(torch.randn((B, N, C), device="cuda").bfloat16(), "seq", True)
    tk = ops.token_keys(keys, B, N)
    print(f"dropout C={C} per sample {clock(lambda: ops.dropout(x, 0.1, keys)):9.0f} us   [{_lib.last_kernel()}]")
    print(f"dropout C={C} per token  {clock(lambda: ops.dropout(x, 0.1, tk, per_row=True)):9.0f} us   [{_lib.last_kernel()}]")
t0 = time.perf_counter(); ops.token_keys(keys, B, N); print(f"token_keys host {1e6 * (time.perf_counter() - t0):.0f} us")
lin1 = eqv.nn.Linear(768, 3072, key=eqv.random.PRNGKey(2))
lin2 = eqv.nn.Linear(3072, 768, key=eqv.random.PRNGKey(3))
x = Act(torch.randn((B, N, 768), device="cuda").bfloat16(), "seq", True)
h = Act(torch.randn((B, N, 3072), device="cuda").bfloat16(), "seq", True)
r = Act(torch.randn((B, N, 768), device="cuda"), "seq", True)
print(f"fc1 + gelu            {clock(lambda: ops.linear(x, lin1, act='gelu')):9.0f} us   [{_lib.last_kernel()}]")
print(f"fc2 + fp32 residual   {clock(lambda: ops.linear(h, lin2, residual=r)):9.0f} us   [{_lib.last_kernel()}]")
print(f"fc2                   {clock(lambda: ops.linear(h, lin2)):9.0f} us   [{_lib.last_kernel()}]")
hd = ops.dropout(h, 0.1, keys)
print(f"fc2 of a dropped h    {clock(lambda: ops.linear(hd, lin2)):9.0f} us   [{_lib.last_kernel()}]")
