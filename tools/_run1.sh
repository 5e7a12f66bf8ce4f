cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for minc in 128 64 32; do
for m in resnet50 swin_t alexnet; do
    echo -n "minc $minc $m fp32: "; MINC=$minc timeout 600 python - <<PY 2>/dev/null
import os, sys, time, torch
sys.path.insert(0, os.getcwd())
import eqxvision_amd as eqv
from eqxvision_amd import _lib
from bench import build_model
_lib.set_flag("f32_lds_minc", int(os.environ["MINC"]))
eqv.set_compute_dtype("fp32")
B = 64 if "$m" != "swin_t" else 32
net = build_model("$m")
x = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False, lanes=1)
for _ in range(3): f(net, x, keys)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(5): f(net, x, keys)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
print(f"{B / dt:.0f} img/s ({dt * 1e3:.1f} ms per {B} images)")
PY
done
done | tee gpurun_out/fp32_minc.txt
