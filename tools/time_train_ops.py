"""GB/s of the training-mode kernels: per-channel batch moments (mv_channel_moments_fwd) and Dropout (mv_dropout_fwd).
usage: time_train_ops.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from eqxvision_amd import _lib as L
s = torch.cuda.current_stream().cuda_stream

def t(fn, n=20):
    for _ in range(3): fn()
    e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3

for rows, C in ((128 * 112 * 112, 64), (128 * 56 * 56, 256), (128 * 28 * 28, 512), (128 * 14 * 14, 1024), (128 * 7 * 7, 2048)):
    x = torch.randn(rows, C, device="cuda").bfloat16()
    ws = torch.empty(int(L.load().mv_channel_moments_ws(C)), device="cuda")
    out = torch.empty(C, device="cuda"); sh = torch.zeros(C, device="cuda")
    for sq in (0, 1):
        us = t(lambda: L.call("mv_channel_moments_fwd", x.data_ptr(), sh.data_ptr() if sq else None, out.data_ptr(), ws.data_ptr(), rows, C, sq, 1, s))
        print(f"moments rows={rows} C={C} squared={sq}: {us:.1f} us  {rows*C*2/us/1e3:.0f} GB/s")
for B, per, C, chw in ((256, 9216, 9216, 0), (128, 56 * 56 * 256, 256, 1), (256, 197 * 768, 768, 0)):
    x = torch.randn(B, per, device="cuda").bfloat16(); y = torch.empty_like(x)
    keys = torch.randint(0, 2**31 - 1, (B, 2), device="cuda", dtype=torch.int32)
    us = t(lambda: L.call("mv_dropout_fwd", x.data_ptr(), keys.data_ptr(), y.data_ptr(), B, per, C, chw, 0.5, 1, s))
    print(f"dropout B={B} per={per} chw={chw}: {us:.1f} us  {B*per*4/us/1e3:.0f} GB/s")
for rows, C in ((128 * 56 * 56, 256), (128 * 112 * 112, 64)):
    x = torch.randn(rows, C, device="cuda").bfloat16(); y = torch.empty_like(x); r = torch.randn(rows, C, device="cuda").bfloat16()
    sc, sh = torch.rand(C, device="cuda") + 0.5, torch.randn(C, device="cuda")
    for flag in (0, 1):
        L.set_flag("affine_scalar", flag)
        us = t(lambda: L.call("mv_channel_affine_fwd", x.data_ptr(), sc.data_ptr(), sh.data_ptr(), y.data_ptr(), rows, C, 1, 1, s))
        print(f"channel_affine rows={rows} C={C} scalar={flag} [{L.last_kernel()}]: {us:.1f} us  {rows*C*4/us/1e3:.0f} GB/s")
    L.set_flag("affine_scalar", 0)
    us = t(lambda: L.call("mv_channel_affine_res_fwd", x.data_ptr(), sc.data_ptr(), sh.data_ptr(), r.data_ptr(), y.data_ptr(), rows, C, 1, 1, s))
    print(f"channel_affine_res rows={rows} C={C}: {us:.1f} us  {rows*C*6/us/1e3:.0f} GB/s")
