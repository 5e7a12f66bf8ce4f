"""Throughput + per-launch worksheet of the segmentation models (section 8 f2) on synthetic data.
usage: seg_bench.py fcn|deeplabv3 [BATCH] [SIZE]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from bench import layer_table
kind = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 32
size = int(sys.argv[3]) if len(sys.argv) > 3 else 224
eqv.set_compute_dtype("bf16")
build = eqv.models.fcn if kind == "fcn" else eqv.models.deeplabv3
net = build(intermediate_layers=lambda m: [m.layer3, m.layer4], aux_in_channels=1024)
net = eqv.tree_inference(eqv.utils.randomize_batchnorm(net, 1), True)
x = torch.rand((B, 3, size, size), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
LANES = int(os.environ.get("LANES", "2"))
f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False, lanes=LANES)
for _ in range(4):
    aux, out = f(net, x, keys)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
e0.record()
for _ in range(20):
    f(net, x, keys)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print(f"{kind}_resnet50 B={B} {size}px lanes={LANES}: {ms:.3f} ms/step  {B / ms * 1e3:.0f} img/s  out {tuple(out.shape)} aux {tuple(aux.shape)}")
rows = layer_table(f._entries()[0], f"gpurun_out/{kind}_layers.txt")
rows.sort(key=lambda r: -r["us"])
for r in rows[:12]:
    print(f"  {r['kernel']:34s} {r['shape']:40s} {r['us']:9.1f} us {r['tflops']:8.1f} TF/s {r['gbs']:8.1f} GB/s")
