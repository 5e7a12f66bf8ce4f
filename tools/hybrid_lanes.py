"""Experiment: two lanes (half batches) for the FIRST k launches of the forward, then the lanes join and the rest of the network
runs as one full-batch launch list -- small late layers (ResNet layer 3 / 4 at 128 images: 98-196 tiles, 128 workgroups) fill the
chip only at the full batch.  Timing only (the joined tail reads the one-lane recording's own buffers).
usage: hybrid_lanes.py MODEL [BATCH]"""
import sys, os, time, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import eqxvision_amd as eqv
from eqxvision_amd import _lib
from bench import build_model

model = sys.argv[1]
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
eqv.set_compute_dtype("bf16")
net = build_model(model)
images = torch.rand((B, 3, 224, 224), dtype=torch.float32).cuda()
keys = eqv.random.split(eqv.random.PRNGKey(0), B)
mk = lambda lanes: eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k), use_graph=lanes > 1, clone_outputs=False, lanes=lanes)
f2, f1 = mk(2), mk(1)
for _ in range(3):
    f2(net, images, keys); f1(net, images, keys)
torch.cuda.synchronize()
c2, c1 = f2._entries()[0], f1._entries()[0]
L0, L1, J = c2.lane_calls[0], c2.lane_calls[1], c1.calls
print([len(L0), len(L1), len(J)])
names = lambda l: [n for _, _, n in l]
same = names(L0) == names(J)
print("same call sequence:", same)
if not same:
    for i, (a, b) in enumerate(zip(names(L0), names(J))):
        print(i, a, b)


def capture(k):
    side = torch.cuda.Stream()
    g = ctypes.c_void_p()
    keep = []
    with torch.cuda.stream(side):
        _lib.call("mv_graph_begin_capture", side.cuda_stream)
        try:
            if k > 0:
                fork = torch.cuda.Event(); fork.record(side)
                bs = torch.cuda.Stream(); bs.wait_event(fork)
                for cfn, args, name in L1[:k]:
                    assert cfn(*args[:-1], bs.cuda_stream) == 0, name
                done = torch.cuda.Event(); done.record(bs)
                for cfn, args, name in L0[:k]:
                    assert cfn(*args[:-1], side.cuda_stream) == 0, name
                side.wait_event(done)
                keep.append((bs, done))
            for cfn, args, name in J[k:]:
                assert cfn(*args[:-1], side.cuda_stream) == 0, name
        finally:
            _lib.call("mv_graph_end_capture", side.cuda_stream, ctypes.byref(g))
    return g, side, keep


ks = [int(a) for a in sys.argv[3].split(",")] if len(sys.argv) > 3 else list(range(0, len(J) + 1, max(1, len(J) // 10))) + [len(J)]
for k in sorted(set(ks)):
    g, side, keep = capture(k)
    best = 1e9
    for rep in range(3):
        for _ in range(5):
            _lib.call("mv_graph_launch", g, side.cuda_stream)
        side.synchronize()
        t0 = time.perf_counter()
        for _ in range(30):
            _lib.call("mv_graph_launch", g, side.cuda_stream)
        side.synchronize()
        best = min(best, (time.perf_counter() - t0) / 30 * 1e3)
    print(f"{model} B={B}: lanes for the first {k:3d} of {len(J)} launches (next: {J[k][2] if k < len(J) else '-'}), joined after: {best:.3f} ms/step  {B / best * 1e3:.0f} img/s", flush=True)
    _lib.call("mv_graph_destroy", g)
