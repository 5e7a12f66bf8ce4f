"""N>1 path on CPU: world_size-2 gloo processes exercise the batch sharding + logits all-gather
(`eqxvision_amd.dist`), with a stand-in forward (the HIP forward itself needs a GPU)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, batch, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from eqxvision_amd import dist as D
    r, w, _ = D.init_from_env("gloo")
    imgs = torch.arange(batch * 3 * 4 * 4, dtype=torch.float32).reshape(batch, 3, 4, 4)

    def forward(x):      # deterministic per-sample "logits"
        return torch.stack([x.sum((1, 2, 3)), x.mean((1, 2, 3)), x.amax((1, 2, 3))], 1)

    out = D.sharded_forward(forward, imgs)
    q.put((rank, out.numpy()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 7])
def test_sharded_forward_gloo(batch):
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker, args=(r, world, port, batch, q)) for r in range(world)]
    [p.start() for p in ps]
    res = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in ps]
    imgs = torch.arange(batch * 3 * 4 * 4, dtype=torch.float32).reshape(batch, 3, 4, 4)
    ref = torch.stack([imgs.sum((1, 2, 3)), imgs.mean((1, 2, 3)), imgs.amax((1, 2, 3))], 1).numpy()
    for r in range(world):
        assert res[r].shape == ref.shape
        np.testing.assert_array_equal(res[r], ref)


def _worker8(rank, world, port, q):
    """BASELINE config 4's collective shape on CPU: 8 ranks x 256 samples, fp32 logits [256, 1000] per rank -> [2048, 1000]."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(1)
    from eqxvision_amd import dist as D
    D.init_from_env("gloo")
    B, classes = 2048, 1000
    imgs = torch.arange(B, dtype=torch.float32).reshape(B, 1, 1, 1).expand(B, 3, 2, 2)      # sample i carries the value i

    def forward(x):      # "logits" of sample i: i + class / 1000 (exact in fp32 for i < 2048): every row identifies its sample
        assert x.shape[0] == B // world
        return x[:, 0, 0, 0, None] + torch.arange(classes, dtype=torch.float32)[None, :] / 1024.0

    out = D.sharded_forward(forward, imgs)
    ok = out.shape == (B, classes) and bool((out[:, 0] == torch.arange(B, dtype=torch.float32)).all()) and \
        bool((out[:, 999] == torch.arange(B, dtype=torch.float32) + 999.0 / 1024.0).all())
    q.put((rank, ok, tuple(out.shape)))
    torch.distributed.destroy_process_group()


def test_sharded_forward_gloo_world8_config4():
    """resnet50 batch 2048 over 8 ranks (BASELINE.json configs[3]): shard bounds, per-rank batch 256 and the all-gather of
    f32[256, 1000] per rank into f32[2048, 1000] in rank order -- the same dist.py control flow the 8-GPU run takes, with gloo
    standing in for RCCL."""
    world, port = 8, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_worker8, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=300) for _ in range(world)]
    [p.join(60) for p in ps]
    assert sorted(r for r, _, _ in res) == list(range(world))
    assert all(ok and shape == (2048, 1000) for _, ok, shape in res), res


def test_shard_bounds_cover_batch():
    from eqxvision_amd.dist import shard_bounds
    for B in (1, 7, 8, 2048):
        for W in (1, 2, 4, 8):
            spans = [shard_bounds(B, r, W) for r in range(W)]
            assert spans[0][0] == 0 and spans[-1][1] == B
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1


def _bn_worker(rank, world, port, batch, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    from eqxvision_amd import dist as D
    D.init_from_env("gloo")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(batch, 5, 6, 8, generator=g) * 3 + 1                 # NHWC map, the same on every rank
    lo, hi = D.shard_bounds(batch, rank, world)
    rows = x[lo:hi].reshape(-1, 8)
    # the two passes of ops.bn_train_update with the device sums replaced by torch (the HIP kernels need a GPU): sum -> all-reduce
    # -> mean; squared deviations from the GLOBAL mean -> all-reduce -> variance
    n = torch.tensor([float(rows.shape[0])])
    D.all_reduce_sum_(n)
    s = rows.sum(0)
    D.all_reduce_sum_(s)
    mean = s / n
    sq = ((rows - mean) ** 2).sum(0)
    D.all_reduce_sum_(sq)
    q.put((rank, D.world_size(), mean.numpy(), (sq / n).numpy()))
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("batch", [8, 7])
def test_batchnorm_moments_all_reduce_gloo(batch):
    """Training-mode BatchNorm's cross-rank moments: every rank ends with the statistics of the whole batch,
    equal shards or ragged."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    ps = [ctx.Process(target=_bn_worker, args=(r, world, port, batch, q)) for r in range(world)]
    [p.start() for p in ps]
    res = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in ps]
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(batch, 5, 6, 8, generator=g) * 3 + 1).reshape(-1, 8).double()
    for rank, w, mean, var in res:
        assert w == 2
        np.testing.assert_allclose(mean, x.mean(0).numpy(), rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(var, x.var(0, unbiased=False).numpy(), rtol=1e-4, atol=1e-5)
