"""HBM microbenchmarks with torch ops (reference points for the memory-bound layers)."""
import torch, time
def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); s = torch.cuda.Event(True); e = torch.cuda.Event(True); s.record()
    for _ in range(n): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) / n * 1e3
for mb in (103, 411, 925):
    n = mb * 1000 * 1000 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda").normal_()
    y = torch.empty_like(x)
    us = t(lambda: y.fill_(1.0)); print(f"fill  {mb} MB: {us:8.1f} us  {mb/us*1e-3*1e3:6.2f} GB/ms = {mb*1e6/us/1e6:6.2f} TB/s(write)")
    us = t(lambda: y.copy_(x)); print(f"copy  {mb} MB: {us:8.1f} us  total {2*mb*1e6/us/1e6:6.2f} TB/s")
    us = t(lambda: x.sum()); print(f"sum   {mb} MB: {us:8.1f} us  {mb*1e6/us/1e6:6.2f} TB/s(read)")
    us = t(lambda: torch.add(x, x, out=y)); print(f"add   {mb} MB: {us:8.1f} us  total {2*mb*1e6/us/1e6:6.2f} TB/s (1R+1W)")
    z = torch.empty_like(x)
    us = t(lambda: torch.add(x, z, out=y)); print(f"add2  {mb} MB: {us:8.1f} us  total {3*mb*1e6/us/1e6:6.2f} TB/s (2R+1W)")
