"""A non-Python host drives a whole network through the C ABI (examples/c_host/alexnet_host.cpp: hipMalloc'ed buffers, a fixed
sequence of mv_* calls captured into a hipGraph through mv_graph_*): SURVEY.md section 8(b)'s "whole-network executor callable from
C".  CPU part: the example compiles and links against the in-tree library.  GPU part: its logits match the oracle and the Python host."""
import os
import shutil
import subprocess
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "examples", "c_host", "alexnet_host.cpp")
LIBDIR = os.path.join(ROOT, "eqxvision_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")


def _build(out):
    if not os.path.exists(os.path.join(LIBDIR, "libeqxvision_amd.so")):
        from eqxvision_amd.build import build
        build(verbose=False)
    cmd = [HIPCC, "-O2", "--offload-arch=gfx950", SRC, "-I" + os.path.join(ROOT, "include"), "-L" + LIBDIR, "-leqxvision_amd",
           "-Wl,-rpath," + LIBDIR, "-o", out]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.skipif(shutil.which(HIPCC) is None and not os.path.exists(HIPCC), reason="hipcc not found")
def test_c_host_example_builds():
    with tempfile.TemporaryDirectory() as d:
        exe = os.path.join(d, "alexnet_host")
        _build(exe)
        assert os.path.getsize(exe) > 0


def _bf16_bytes(a):
    from oracle import np_ops as O
    r = O.bf16_round(np.asarray(a, np.float32))
    return (r.view(np.uint32) >> 16).astype(np.uint16).tobytes()


@pytest.mark.gpu
def test_c_host_runs_alexnet_through_the_abi():
    import torch
    assert torch.cuda.is_available()
    from oracle import state as S
    from oracle import torch_ref as TR
    B, classes = 4, 1000
    sd = S.alexnet_state(1, classes)
    x = S.synthetic_images(B, 224, seed=2)
    krsc = lambda w: np.ascontiguousarray(np.asarray(w, np.float32).transpose(0, 2, 3, 1))
    blobs = [x.astype(np.float32).tobytes(), _bf16_bytes(sd["features.0.weight"]), np.asarray(sd["features.0.bias"], np.float32).tobytes()]
    for i in (3, 6, 8, 10):
        blobs += [_bf16_bytes(krsc(sd[f"features.{i}.weight"])), np.asarray(sd[f"features.{i}.bias"], np.float32).tobytes()]
    for i in (1, 4, 6):
        blobs += [_bf16_bytes(sd[f"classifier.{i}.weight"]), np.asarray(sd[f"classifier.{i}.bias"], np.float32).tobytes()]
    ref = TR.alexnet_forward(sd, x).numpy()
    with tempfile.TemporaryDirectory() as d:
        exe, wfile, ofile = (os.path.join(d, n) for n in ("alexnet_host", "w.bin", "o.bin"))
        _build(exe)
        with open(wfile, "wb") as f:
            f.write(np.asarray([B, classes], np.int32).tobytes())
            for b in blobs:
                f.write(np.asarray([len(b)], np.int64).tobytes())
                f.write(b)
        r = subprocess.run([exe, wfile, ofile], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
        got = np.fromfile(ofile, np.float32).reshape(B, classes)
    err = float(np.abs(got - ref).max())
    assert np.isfinite(got).all() and err <= 1e-2 * max(1.0, float(np.abs(ref).max())), (err, r.stdout)
    # (no arg-max assertion: the synthetic checkpoint's top logits sit closer together than the bf16 tolerance)
