"""Experiment (round 5): the two lanes on DISJOINT halves of the chip (hipExtStreamCreateWithCUMask, 16 CUs of every XCD each) AND
out of phase -- lane 1 starts when lane 0 has reached call k of its launch list, both free-running from then on -- so that one
lane's HBM-bound part of the network (tools/ubench/cu_bw.hip: 128 CUs pull 5.25 of the 5.9 TB/s all 256 reach) overlaps the other
lane's matrix-bound part (which clocks higher on half the chip).   usage: masked_offset_lanes.py MODEL [BATCH]"""
import sys, os, time, ctypes
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
    ref = f(net, images, keys)
torch.cuda.synchronize()
ref = ref.clone()
t0 = time.perf_counter()
for _ in range(20):
    f(net, images, keys)
torch.cuda.synchronize()
g = (time.perf_counter() - t0) / 20 * 1e3
print(f"{model} B={B}: graph, two lanes joined per step: {g:.3f} ms/step  {B / g * 1e3:.0f} img/s", flush=True)
c = f._entries()[0]
lanes = c.lane_calls
n = len(lanes[0])
hip = ctypes.CDLL("libamdhip64.so")
hip.hipExtStreamCreateWithCUMask.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint32, ctypes.POINTER(ctypes.c_uint32)]
hip.hipStreamCreateWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_uint]
hip.hipStreamWaitEvent.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_uint]
hip.hipEventCreate.argtypes = [ctypes.POINTER(ctypes.c_void_p)]
hip.hipEventRecord.argtypes = [ctypes.c_void_p, ctypes.c_void_p]


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    assert hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr) == 0
    return s.value


def plain_stream():
    s = ctypes.c_void_p()
    assert hip.hipStreamCreateWithFlags(ctypes.byref(s), 1) == 0
    return s.value


def run(streams, k, steps=20):
    """lane 1's call j is issued together with lane 0's call j + k; in step 0 lane 1 also WAITS for lane 0's call k - 1."""
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for st in range(steps):
            for i in range(n + k):
                if i < n:
                    cfn, args, name = lanes[0][i]
                    assert cfn(*args[:-1], streams[0]) == 0, name
                    if st == 0 and k and i == k - 1:
                        ev = ctypes.c_void_p()
                        hip.hipEventCreate(ctypes.byref(ev)); hip.hipEventRecord(ev, streams[0]); hip.hipStreamWaitEvent(streams[1], ev, 0)
                j = i - k
                if 0 <= j < len(lanes[1]):
                    cfn, args, name = lanes[1][j]
                    assert cfn(*args[:-1], streams[1]) == 0, name
        torch.cuda.synchronize()
        best = min(best, (time.perf_counter() - t0) / steps * 1e3)
    err = (c.out.float() - ref.float()).abs().max().item()
    return best, err


# XCD-interleaved CU numbering of the mask words is not documented here: tools/ubench/cu_mask_census.hip showed that word w of the
# mask covers 32 consecutive "logical" CUs that the runtime spreads over the XCDs; half the words = half the CUs of every XCD
half = [[0xffffffff if w < 4 else 0 for w in range(8)], [0xffffffff if w >= 4 else 0 for w in range(8)]]
alt = [[0x55555555] * 8, [0xaaaaaaaa] * 8]
for tag, mk in (("plain streams       ", None), ("masks words 0-3/4-7", half), ("masks even/odd bits", alt)):
    st = [plain_stream(), plain_stream()] if mk is None else [masked_stream(m) for m in mk]
    for k in sorted({0, n // 5, n // 3, (2 * n) // 5, n // 2, (3 * n) // 5, (2 * n) // 3}):
        t, err = run(st, k)
        print(f"  {tag} lane 1 starts at call {k:3d}/{n}: {t:.3f} ms/step  {B / t * 1e3:7.0f} img/s   (max |diff| vs graph {err:.1e})", flush=True)
