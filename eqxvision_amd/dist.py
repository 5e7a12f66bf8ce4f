"""Multi-GPU driver for the batch-sharded forward.

The reference has no distributed code: samples are independent (`jax.vmap`, README.md:37-40) and the
only cross-sample op, BatchNorm batch statistics, is inactive in inference.  So the path shards
naturally: one process per GPU, weights replicated, rank r takes `images[r*B/W:(r+1)*B/W]`, and ONE
all-gather of the fp32 logits (`[B/W, classes]`, ~1 MB per rank -> latency-bound, a direct exchange over the
7 xGMI links) rebuilds the `(B, classes)` result the single-device vmap would return.  No data-path
collective before that.

The collective itself is the library's `mv_allgather` (RCCL behind the C ABI, `csrc/comm.hip`), enqueued on
the same stream as the kernels.  `torch.distributed` is only the bootstrap: a gloo process group ships the
128-byte RCCL unique id from rank 0 and provides the host-side barrier / max-over-ranks of the benchmark.
`EQV_DIST_COLLECTIVE=torch` routes the all-gather through `torch.distributed` instead (backend nccl == RCCL on
ROCm, or gloo with host staging) -- that is also what ragged batches and the CPU tests use.
"""
from __future__ import annotations

import ctypes
import os
from typing import Callable, Optional, Tuple

import torch
import torch.distributed as dist

from . import _lib

_state = {"native": False, "group": None, "collective": None}   # collective: what init_from_env set out to use ("rccl" / "torch")


def _rccl_path() -> Optional[str]:
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


def native_comm_init(rank: int, world: int) -> None:
    """mv_comm_init over the initialised torch.distributed group (any backend): rank 0 draws the RCCL unique id,
    everybody receives it, every rank binds its CURRENT HIP device."""
    if "EQV_RCCL_LIB" not in os.environ and _rccl_path():
        os.environ["EQV_RCCL_LIB"] = _rccl_path()      # the copy PyTorch already mapped: one RCCL per process
    box = [None]
    if rank == 0:
        try:
            buf = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
            _lib.call("mv_comm_unique_id", buf, _lib.COMM_ID_BYTES)
            box[0] = bytes(buf.raw)
        except _lib.MVError as e:          # tell the other ranks instead of leaving them in the broadcast
            box[0] = f"ERR {e}"
    if world > 1:
        dist.broadcast_object_list(box, src=0)
    if not isinstance(box[0], bytes):
        raise _lib.MVError(f"rank 0 could not create the RCCL unique id: {box[0]}")
    uid = ctypes.create_string_buffer(box[0], _lib.COMM_ID_BYTES)
    _lib.call("mv_comm_init", rank, world, uid)
    _state["native"] = True


def native_comm_destroy() -> None:
    if _state["native"]:
        _lib.call("mv_comm_destroy")
        _state["native"] = False


def init_from_env(backend: Optional[str] = None) -> Tuple[int, int, int]:
    """Initialise from RANK / WORLD_SIZE / LOCAL_RANK / MASTER_* (torchrun contract).  Returns (rank, world, local_rank);
    a no-op single-process setup when WORLD_SIZE is unset/1."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    # test hooks (a 1-GPU box cannot host two RCCL ranks): EQV_DIST_BACKEND=gloo, EQV_DIST_DEVICE=0 put every rank on
    # one device with a host-staged collective so the multi-rank control flow can be exercised with the real HIP forward
    backend = backend or os.environ.get("EQV_DIST_BACKEND")
    if "EQV_DIST_DEVICE" in os.environ:
        local = int(os.environ["EQV_DIST_DEVICE"])
    has_gpu = torch.cuda.is_available()
    collective = os.environ.get("EQV_DIST_COLLECTIVE") or ("rccl" if has_gpu and backend is None else "torch")
    _state["collective"] = collective
    if has_gpu:
        torch.cuda.set_device(local)
    if world > 1 and not dist.is_initialized():
        if backend is None:
            backend = "gloo" if (collective == "rccl" or not has_gpu) else "nccl"
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    if world > 1 and collective == "rccl" and not _state["native"]:
        try:
            native_comm_init(rank, world)
        except _lib.MVError as e:
            # The library's own RCCL binding could not come up (e.g. librccl not loadable).  DESIGN.md section 6 promises ONE
            # collective, `mv_allgather` on the launch stream; a different backend is never chosen silently.  Opt in with
            # EQV_DIST_ALLOW_TORCH=1 to keep a run alive on PyTorch's binding of the same library (a second process group,
            # backend nccl == RCCL; never a host-staged gather).
            if os.environ.get("EQV_DIST_ALLOW_TORCH") != "1":
                raise _lib.MVError(
                    f"mv_comm_init failed ({e}); the sharded forward needs the native RCCL communicator. "
                    "Set EQV_DIST_ALLOW_TORCH=1 to route the logits all-gather through torch.distributed (nccl) instead.") from e
            import warnings
            warnings.warn(f"mv_comm_init failed ({e}); EQV_DIST_ALLOW_TORCH=1: the logits all-gather goes through torch.distributed (nccl)")
            _state["group"] = dist.new_group(backend="nccl")
    return rank, world, local


def barrier() -> None:
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float) -> float:
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return value
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([value], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def shard_bounds(batch: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of `batch` samples: the first batch%world ranks take one extra."""
    base, extra = divmod(batch, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


def shard(images, rank: int, world: int):
    lo, hi = shard_bounds(images.shape[0], rank, world)
    return images[lo:hi]


def _torch_all_gather(out: torch.Tensor, local: torch.Tensor, group=None) -> None:
    if local.is_cuda and dist.get_backend(group) == "gloo":     # gloo has no device all-gather: stage through the host
        h = torch.empty(out.shape, dtype=out.dtype)
        dist.all_gather_into_tensor(h, local.cpu().contiguous(), group=group)
        out.copy_(h)
    else:
        dist.all_gather_into_tensor(out, local.contiguous(), group=group)


def all_gather_rows(local: torch.Tensor, batch: int, group=None, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Gather per-rank `[b_r, ...]` row blocks into `[batch, ...]` on every rank (rank order)."""
    native = _state["native"] and group is None and local.is_cuda
    if group is None and not native and _state["group"] is not None:
        group = _state["group"]
    if native:
        world = _lib.load().mv_comm_size()
    elif dist.is_initialized():
        world = dist.get_world_size(group)
    else:
        return local
    if world == 1 and not native:
        return local
    if batch % world == 0:          # equal shards: ONE all-gather call
        if out is None:
            out = local.new_empty((batch,) + tuple(local.shape[1:]))
        local = local.contiguous()
        if native:
            _lib.call("mv_allgather", local.data_ptr(), out.data_ptr(), local.numel() * local.element_size(),
                      torch.cuda.current_stream().cuda_stream)
            out._eqv_src = local            # keep the send buffer alive until the stream is done with it
        else:
            _torch_all_gather(out, local, group)
        return out
    # ragged: pad every shard to the largest, gather, then drop the padding
    per = [shard_bounds(batch, r, world) for r in range(world)]
    mx = max(hi - lo for lo, hi in per)
    pad = local.new_zeros((mx,) + tuple(local.shape[1:]))
    pad[: local.shape[0]] = local
    full = local.new_empty((world * mx,) + tuple(local.shape[1:]))
    if native:
        _lib.call("mv_allgather", pad.data_ptr(), full.data_ptr(), pad.numel() * pad.element_size(),
                  torch.cuda.current_stream().cuda_stream)
    else:
        _torch_all_gather(full, pad, group)
    return torch.cat([full[r * mx: r * mx + (hi - lo)] for r, (lo, hi) in enumerate(per)], 0)


def all_reduce_sum_(t: torch.Tensor, group=None) -> torch.Tensor:
    """In-place sum over ranks of a small fp32 tensor (training-mode BatchNorm moments: 2 x C floats per layer).  RCCL
    (`mv_allreduce_sum_f32`) when the native communicator is up, else `torch.distributed` (gloo stages through the host);
    a single process returns `t` unchanged."""
    native = _state["native"] and group is None and t.is_cuda
    if group is None and not native and _state["group"] is not None:
        group = _state["group"]
    if native:
        if _lib.load().mv_comm_size() > 1:
            _lib.call("mv_allreduce_sum_f32", t.data_ptr(), t.numel(), torch.cuda.current_stream().cuda_stream)
        return t
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return t
    if t.is_cuda and dist.get_backend(group) == "gloo":
        h = t.cpu()
        dist.all_reduce(h, op=dist.ReduceOp.SUM, group=group)
        t.copy_(h)
    else:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return t


def world_size(group=None) -> int:
    if _state["native"] and group is None:
        return max(1, _lib.load().mv_comm_size())
    if group is None and _state["group"] is not None:
        group = _state["group"]
    return dist.get_world_size(group) if dist.is_initialized() else 1


def sharded_forward(forward: Callable, images, *, global_batch: Optional[int] = None, group=None) -> torch.Tensor:
    """Run `forward(local_images) -> [b_local, classes]` on this rank's shard of the GLOBAL batch
    `images` and all-gather the logits."""
    if not dist.is_initialized() and not _state["native"]:
        return forward(images)
    if dist.is_initialized():
        rank, world = dist.get_rank(group), dist.get_world_size(group)
    else:
        rank, world = _lib.load().mv_comm_rank(), _lib.load().mv_comm_size()
    B = images.shape[0] if global_batch is None else global_batch
    local = forward(shard(images, rank, world))
    return all_gather_rows(local, B, group)
