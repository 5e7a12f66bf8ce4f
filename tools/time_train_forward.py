"""ms per TRAINING-MODE forward (BatchNorm on batch statistics, eager launches): usage: time_train_forward.py [model] [B]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from bench import build_model
name = sys.argv[1] if len(sys.argv) > 1 else "resnet50"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eqv.set_compute_dtype("bf16")
net = eqv.tree_inference(build_model(name), False)
x = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
f = lambda: eqv.vmap(net, axis_name="batch")(x, key=keys)
for _ in range(3): f()
torch.cuda.synchronize(); t0 = time.perf_counter()
n = 10
for _ in range(n): f()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
print(f"{name} B={B} training-mode forward: {dt*1e3:.2f} ms/step  {B/dt:.0f} img/s")
