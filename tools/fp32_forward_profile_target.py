import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import eqxvision_amd as eqv
from eqxvision_amd import _lib
from bench import build_model
m = sys.argv[1]
eqv.set_compute_dtype("fp32")
B = 64 if m != "swin_t" else 32
net = build_model(m)
x = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=False, clone_outputs=False, lanes=1)
for _ in range(4): f(net, x, keys)
torch.cuda.synchronize()
