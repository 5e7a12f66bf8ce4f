"""Experiment: run the two lanes of the batch on DISJOINT halves of the chip (hipExtStreamCreateWithCUMask: 16 CUs of every
XCD each) instead of letting two graph branches time-slice all 256 CUs.  Eager two-stream replay of the recorded launch
lists, free-running (no per-step join), so that one lane's HBM-bound kernels can overlap the other's MFMA-bound ones.
usage: cu_mask_lanes.py MODEL [BATCH] [LANES]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from eqxvision_amd import _lib
from bench import build_model

model = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
NL = int(sys.argv[3]) if len(sys.argv) > 3 else 2
eqv.set_compute_dtype("bf16")
net = build_model(model)
images = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
f = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=True, clone_outputs=False, lanes=NL)
for _ in range(4):
    ref = f(net, images, keys)
torch.cuda.synchronize()
ref = ref.clone()

def timed(fn, steps=20, reps=3):
    out = []
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn(steps)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) / steps * 1e3)
    return min(out), out

def graph_steps(n):
    for _ in range(n):
        f(net, images, keys)
g, allg = timed(graph_steps)
print(f"{model} B={B} lanes={NL}: graph (time-sliced lanes) {g:.3f} ms/step {B/g*1e3:.0f} img/s  {['%.3f' % x for x in allg]}", flush=True)

c = f._entries()[0]
lane_calls = c.lane_calls
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]

def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return s.value

def plain_stream():
    s = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithFlags(ctypes.byref(s), 1)
    assert rc == 0, rc
    return s.value

def lane_masks(nl):
    per = 8 // nl
    return [[0xffffffff if l * per <= w < (l + 1) * per else 0 for w in range(8)] for l in range(nl)]

def eager(streams, offset=0):
    """issue the lanes' launches interleaved call by call from this thread; lane l runs `offset*l` calls ahead ... the
    streams free-run across steps."""
    n = max(len(lc) for lc in lane_calls)
    def run(steps):
        for _ in range(steps):
            for i in range(n + offset * (len(lane_calls) - 1)):
                for l, lc in enumerate(lane_calls):
                    j = i - offset * l
                    if 0 <= j < len(lc):
                        cfn, args, name = lc[j]
                        rc = cfn(*args[:-1], streams[l])
                        assert rc == 0, name
    return run

def check(tag):
    torch.cuda.synchronize()
    err = (c.out.float() - ref.float()).abs().max().item()
    print(f"    {tag}: max |diff| vs graph result {err:.3e}", flush=True)

plain = [plain_stream() for _ in range(NL)]
t, allt = timed(eager(plain))
print(f"  eager, {NL} plain streams        : {t:.3f} ms/step {B/t*1e3:.0f} img/s  {['%.3f' % x for x in allt]}", flush=True)
check("plain")
masked = [masked_stream(m) for m in lane_masks(NL)]
for off in (0,):
    t, allt = timed(eager(masked, off))
    print(f"  eager, {NL} CU-masked streams off={off:2d}: {t:.3f} ms/step {B/t*1e3:.0f} img/s  {['%.3f' % x for x in allt]}", flush=True)
check("masked")
# different stream priorities for the two lanes: lane 0's kernels get the CUs first, lane 1 fills what is left
hip.hipStreamCreateWithPriority.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint, ctypes.c_int]
lo_p, hi_p = ctypes.c_int(), ctypes.c_int()
hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_p), ctypes.byref(hi_p))
def prio_stream(pr):
    s = ctypes.c_void_p()
    rc = hip.hipStreamCreateWithPriority(ctypes.byref(s), 1, pr)
    assert rc == 0, rc
    return s.value
pst = [prio_stream(hi_p.value if l == 0 else lo_p.value) for l in range(NL)]
t, allt = timed(eager(pst))
print(f"  eager, priorities {hi_p.value} / {lo_p.value}            : {t:.3f} ms/step {B/t*1e3:.0f} img/s  {['%.3f' % x for x in allt]}", flush=True)
check("priorities")
t, allt = timed(eager(plain))
print(f"  eager, {NL} plain streams (again) : {t:.3f} ms/step {B/t*1e3:.0f} img/s  {['%.3f' % x for x in allt]}", flush=True)
# host-side issue rate, for reference
t0 = time.perf_counter(); eager(plain)(3); t1 = time.perf_counter()
print(f"  (host issue time per step, queue not full: {(t1-t0)/3*1e3:.2f} ms)")
