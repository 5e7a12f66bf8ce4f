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


def main():
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
