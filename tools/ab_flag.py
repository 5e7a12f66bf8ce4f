"""A/B a library flag on one bench workload inside ONE process / one GPU box (box-to-box variance is +-3%):
usage: ab_flag.py FLAG MODEL [BATCH] [REPS]  -> ms/step with FLAG=0 and FLAG=1, interleaved."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from eqxvision_amd import _lib
from bench import build_model
flag, model = sys.argv[1], sys.argv[2]
B = int(sys.argv[3]) if len(sys.argv) > 3 else 256
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 3
eqv.set_compute_dtype("bf16")
net = build_model(model)
images = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
fw = {}
for v in (0, 1):
    if flag.startswith("lanes"):
        f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False,
                           lanes=int(os.environ.get('BASE_LANES', '1')) if v == 0 else int(flag[5:] or 2))
    else:
        _lib.set_flag(flag, v * int(os.environ.get('FLAGVAL', '1')))
        f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False,
                           lanes=int(os.environ.get("LANES", "2")))
    for _ in range(4): f(net, images, keys)
    fw[v] = f
torch.cuda.synchronize()
for r in range(reps):
    for v in (0, 1):
        if not flag.startswith("lanes"):
            _lib.set_flag(flag, v * int(os.environ.get('FLAGVAL', '1')))      # the jit cache keys on the switch settings
        e0, e1 = torch.cuda.Event(True), torch.cuda.Event(True)
        e0.record()
        for _ in range(20): fw[v](net, images, keys)
        e1.record(); torch.cuda.synchronize()
        print(f"{flag}={v}: {e0.elapsed_time(e1)/20:.4f} ms/step  {B/(e0.elapsed_time(e1)/20)*1e3:.0f} img/s", flush=True)
if not flag.startswith('lanes'): _lib.set_flag(flag, 0)
