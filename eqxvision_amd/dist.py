"""Multi-GPU driver for the batch-sharded forward.

The reference has no distributed code: samples are independent (`jax.vmap`, README.md:37-40) and the
only cross-sample op, BatchNorm batch statistics, is inactive in inference.  So the path shards
naturally: one process per GPU (`torch.distributed`, backend "nccl" == RCCL over xGMI on ROCm),
weights replicated, rank r takes `images[r*B/W:(r+1)*B/W]`, and ONE all-gather of the fp32 logits
(`[B/W, classes]`, ~1 MB per rank -> latency-bound, a direct exchange over the 7 xGMI links) rebuilds the
`(B, classes)` result the single-device vmap would return.  No data-path collective before that.
"""
from __future__ import annotations

import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise torch.distributed from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).
    Returns (rank, world, local_rank); a no-op single-process setup when WORLD_SIZE is unset/1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # test hooks (a 1-GPU box cannot host two RCCL ranks): EQV_DIST_BACKEND=gloo, EQV_DIST_DEVICE=0 put every rank on
    # one device with a host-staged collective so the multi-rank control flow of bench.py can be exercised
    backend = backend or os.environ.get("EQV_DIST_BACKEND")
    if "EQV_DIST_DEVICE" in os.environ:
        local = int(os.environ["EQV_DIST_DEVICE"])
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if torch.cuda.is_available():
            torch.cuda.set_device(local)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    elif torch.cuda.is_available():
        torch.cuda.set_device(local)
    return rank, world, local


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `batch` samples: the first batch%world ranks take one extra."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(images, rank: int, world: int):
    lo, hi = shard_bounds(images.shape[0], rank, world)
    return images[lo:hi]


def all_gather_rows(local: torch.Tensor, batch: int, group=None) -> torch.Tensor:
    """Gather per-rank `[b_r, ...]` row blocks into `[batch, ...]` on every rank (rank order)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return local
    world = dist.get_world_size(group)
    if batch % world == 0:          # equal shards: one all_gather_into_tensor (a single RCCL call)
        out = local.new_empty((batch,) + tuple(local.shape[1:]))
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)
        return out
    # ragged: pad every shard to the largest, gather, then drop the padding
    per = [shard_bounds(batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in per)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    out = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    dist.all_gather_into_tensor(out, pad, group=group)
    return torch.cat([out[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(per)], 0)


def sharded_forward(forward: Callable, images, *, global_batch: Optional[int] = None, group=None) -> torch.Tensor:
    """Run `forward(local_images) -> [b_local, classes]` on this rank's shard of the GLOBAL batch
    `images` and all-gather the logits."""
    if not dist.is_initialized():
        return forward(images)
    rank, world = dist.get_rank(group), dist.get_world_size(group)
    B = images.shape[0] if global_batch is None else global_batch
    local = forward(shard(images, rank, world))
    return all_gather_rows(local, B, group)
