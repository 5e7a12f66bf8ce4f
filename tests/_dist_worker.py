"""Worker of tests/test_dist_gpu.py: one rank of a 2-rank job on ONE device (EQV_DIST_BACKEND=gloo, EQV_DIST_DEVICE=0).
Runs the REAL HIP forward on this rank's shard, all-gathers the logits through eqxvision_amd.dist and (rank 0) compares
the gathered [B, classes] with the single-rank forward of the whole batch.  usage: _dist_worker.py OUT.json B"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402

import eqxvision_amd as eqv  # noqa: E402
from eqxvision_amd import dist as D  # noqa: E402


def bn_train(out, B):
    """Training-mode BatchNorm over a data-parallel batch: every rank runs resnet18 (BatchNorm NOT in inference mode) on its
    shard; the batch moments are summed over the ranks (eqxvision_amd.dist.all_reduce_sum_), so every rank ends with the SAME
    running statistics, and those of the first BatchNorm equal 0.01 * (moments of conv1 over the WHOLE batch) + 0.99 * old."""
    import torch.distributed as dist
    from oracle import np_ops as O
    rank, world, local = D.init_from_env()
    eqv.set_compute_dtype("bf16")
    base = eqv.utils.randomize_batchnorm(eqv.models.resnet18(key=eqv.random.PRNGKey(1)), 1)
    old_m, old_v = (np.array(a, np.float32) for a in base.bn1.state_index.value)
    net = eqv.tree_inference(base, False)            # the StateIndex slots are shared with `base`
    x = np.random.Generator(np.random.PCG64(0)).random((B, 3, 64, 64), dtype=np.float32)
    keys = eqv.random.split(eqv.random.PRNGKey(0), B)
    lo, hi = D.shard_bounds(B, rank, world)
    got = eqv.vmap(net, axis_name="batch")(x[lo:hi], key=keys[lo:hi])
    torch.cuda.synchronize()
    stats = [np.asarray(a, np.float32) for bn in (net.bn1, net.layer4.layers[1].bn2) for a in bn.state_index.value]
    t = torch.from_numpy(np.concatenate([a.ravel() for a in stats]))
    ts = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(ts, t)
    agree = float(max((a - ts[0]).abs().max() for a in ts))
    if rank == 0:
        bf = lambda a: torch.from_numpy(np.asarray(a, np.float32)).to(torch.bfloat16).to(torch.float32).numpy()
        w = bf(np.asarray(net.conv1.weight))
        ys = bf(np.stack([O.conv2d(bf(im), w, None, 2, 3) for im in x])).astype(np.float64)     # conv1 over the GLOBAL batch
        exp_m = 0.01 * ys.mean(axis=(0, 2, 3)) + 0.99 * old_m
        exp_v = 0.01 * ys.var(axis=(0, 2, 3)) + 0.99 * old_v
        info = {"rank": rank, "world": world, "rank_agreement": agree, "shard_rows": int(hi - lo), "logits_shape": list(got.shape),
                "bn1_mean_err": float(np.abs(stats[0] - exp_m).max()), "bn1_var_err": float(np.abs(stats[1] - exp_v).max())}
        info["ok"] = bool(agree == 0.0 and info["bn1_mean_err"] < 1e-5 and info["bn1_var_err"] < 1e-5)
        json.dump(info, open(out, "w"))
    D.barrier()


def main():
    if len(sys.argv) > 3 and sys.argv[3] == "bn_train":
        return bn_train(sys.argv[1], int(sys.argv[2]))
    out, B = sys.argv[1], int(sys.argv[2])
    rank, world, local = D.init_from_env()
    eqv.set_compute_dtype("bf16")
    key = eqv.random.PRNGKey(1)
    net = eqv.tree_inference(eqv.utils.randomize_batchnorm(eqv.models.resnet50(key=key), 1), True)
    x = np.random.Generator(np.random.PCG64(0)).random((B, 3, 64, 64), dtype=np.float32)
    fwd = eqv.filter_jit(lambda n, im, k: eqv.vmap(n, axis_name="batch")(im, key=k))

    def forward(im):
        return fwd(net, im, eqv.random.split(eqv.random.PRNGKey(0), im.shape[0]))

    got = None
    for _ in range(3):                               # trace, capture, replay: the gather follows every one of them
        got = D.sharded_forward(forward, x)
    torch.cuda.synchronize()
    info = {"rank": rank, "world": world, "shape": list(got.shape)}
    if rank == 0:
        ref = forward(x)                             # the whole batch on one rank (a second signature of the same jit)
        torch.cuda.synchronize()
        d = float((got - ref).abs().max())
        lo, hi = D.shard_bounds(B, 1, world)
        info.update(err=d, refmax=float(ref.abs().max()), ok=bool(d <= 2e-2 * max(1.0, float(ref.abs().max()))),
                    rows_rank1_nonzero=bool(got[lo:hi].abs().sum() > 0))
        json.dump(info, open(out, "w"))
    D.barrier()


if __name__ == "__main__":
    main()
