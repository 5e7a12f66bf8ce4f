"""`-m gpu`: the multi-rank path with the REAL HIP forward (BASELINE config 4's control flow on a 1-GPU box).
A 1-GPU box cannot host two RCCL ranks, so (a) two processes share device 0 and gather through gloo (host staged) --
forward + all_gather_rows + rank-order row placement are the production code; (b) the RCCL entry points of the C ABI
(mv_comm_unique_id / mv_comm_init / mv_allgather / mv_comm_destroy) run with a 1-rank communicator; (c) bench.py --gpus 2
spawns its own ranks and reports n_gpus = 2."""
import json
import os
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _env(port):
    env = dict(os.environ)
    env.update(EQV_DIST_BACKEND="gloo", EQV_DIST_DEVICE="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    return env


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.parametrize("B", [8, 7])        # even and ragged shards
def test_two_ranks_hip_forward_and_gather(tmp_path, B):
    assert torch.cuda.is_available()
    out = tmp_path / "r.json"
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), str(B)]
    r = subprocess.run(cmd, env=_env(port), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.load(open(out))
    assert info["world"] == 2 and info["shape"] == [B, 1000] and info["rows_rank1_nonzero"] and info["ok"], info


def test_two_ranks_training_mode_batchnorm(tmp_path):
    """SURVEY section 8 f4, first item: BatchNorm batch statistics over a data-parallel batch (the reference's `pmean` over
    axis_name, across GPUs): 2 ranks x 8 images = the moments of 16."""
    assert torch.cuda.is_available()
    out = tmp_path / "bn.json"
    port = _free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "tests", "_dist_worker.py"), str(out), "16", "bn_train"]
    r = subprocess.run(cmd, env=_env(port), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    info = json.load(open(out))
    assert info["world"] == 2 and info["shard_rows"] == 8 and info["logits_shape"] == [8, 1000] and info["ok"], info


def test_rccl_abi_single_rank():
    """mv_comm_* over librccl on the device: a 1-rank communicator's all-gather is a copy on the launch stream."""
    from eqxvision_amd import _lib, dist as D
    assert torch.cuda.is_available()
    torch.cuda.set_device(0)
    L = _lib.load()
    assert L.mv_comm_size() == 0
    D.native_comm_init(0, 1)
    try:
        assert L.mv_comm_size() == 1 and L.mv_comm_rank() == 0
        x = torch.randn(256, 1000, device="cuda")
        y = D.all_gather_rows(x, 256)
        torch.cuda.synchronize()
        assert y.data_ptr() != x.data_ptr() and torch.equal(x, y)
        s = torch.randn(2 * 2048, device="cuda")
        s0 = s.clone()
        assert D.all_reduce_sum_(s) is s                        # mv_allreduce_sum_f32 skipped for one rank ...
        _lib.call("mv_allreduce_sum_f32", s.data_ptr(), s.numel(), torch.cuda.current_stream().cuda_stream)   # ... ncclAllReduce itself
        torch.cuda.synchronize()
        assert torch.equal(s, s0)
        with pytest.raises(_lib.MVError):                       # one communicator per process
            import ctypes
            _lib.call("mv_comm_init", 0, 1, ctypes.create_string_buffer(128))
    finally:
        D.native_comm_destroy()
    assert L.mv_comm_size() == 0


def test_bench_spawns_its_ranks():
    port = _free_port()
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--batch", "32",
                        "--soak", "0", "--no-cpu"], env=_env(port), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 2 and line["config"]["global_batch"] == 64 and line["config"]["parallelism"] == "dp2", line
    assert line["value"] > 0


def test_bench_one_rank_logits_through_mv_allgather():
    """Round-4 review, item 7: `bench.py --gpus 1` with EQV_FORCE_COMM=1 pushes the logits through mv_allgather on a 1-rank RCCL
    communicator INSIDE the timed region (bench.py asserts the gathered buffer equals the local logits bit for bit) -- the code
    path of BASELINE configs[3]'s 8-GPU line, executed at N = 1."""
    env = dict(os.environ)
    env.update(EQV_FORCE_COMM="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "5", "--warmup", "2", "--batch", "64",
                        "--soak", "0", "--no-cpu", "--extra", "none", "--no-lanes1"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    assert line["n_gpus"] == 1 and line["config"]["collective"] == "mv_allgather (RCCL)" and line["config"]["collective_forced_1rank"], line
    assert line["value"] > 0
