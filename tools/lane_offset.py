"""Experiment: the two lanes (half batches) of a forward on two free-running streams, lane 1 started when lane 0 has reached a
given fraction of its launch list -- does one lane's MFMA-bound half of the network overlap the other's HBM-bound half?
usage: lane_offset.py MODEL [BATCH]"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from bench import build_model

model = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eqv.set_compute_dtype("bf16")
net = build_model(model)
images = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False, lanes=2)
for _ in range(4):
    f(net, images, keys)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    f(net, images, keys)
torch.cuda.synchronize()
g = (time.perf_counter() - t0) / 20 * 1e3
print(f"{model} B={B}: graph, two lanes joined per step: {g:.3f} ms/step  {B / g * 1e3:.0f} img/s", flush=True)
lanes = f._entries()[0].lane_calls
s0, s1 = torch.cuda.Stream(), torch.cuda.Stream()
n = len(lanes[0])
for frac in (0.0, 0.15, 0.3, 0.5, 0.7):
    k = int(frac * n)
    steps = 20
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in range(steps):
            for i in range(n + k):
                if i < n:
                    cfn, args, name = lanes[0][i]
                    assert cfn(*args[:-1], s0.cuda_stream) == 0, name
                    if st == 0 and i == k - 1:
                        ev = torch.cuda.Event(); ev.record(s0); s1.wait_event(ev)
                j = i - k
                if 0 <= j < len(lanes[1]):
                    cfn, args, name = lanes[1][j]
                    assert cfn(*args[:-1], s1.cuda_stream) == 0, name
        torch.cuda.synchronize()
        t = (time.perf_counter() - t0) / steps * 1e3
    print(f"  free-running lanes, lane 1 starts at call {k}/{n} of lane 0: {t:.3f} ms/step  {B / t * 1e3:.0f} img/s", flush=True)
