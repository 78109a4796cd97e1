"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in CPU tests).  The hot path has exactly one exchange: a SUM all-reduce of the
flat ``[grads | loss]`` buffer per optimiser step (the loss is a row SUM, so the global-batch gradient
is the sum -- not the mean -- of shard gradients).  compress / decompress shard
rows with no collective.
"""
import os

import torch
import torch.distributed as td


def is_dist():
    return td.is_available() and td.is_initialized()


def rank_world():
    if is_dist():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world, local_rank).  No-op for single-process runs."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: several ranks may share one GPU (BALER_AMD_FORCE_DEVICE) over gloo (BALER_AMD_DIST_BACKEND)
    # so that the multi-rank code path can be exercised on a single-GPU box; RCCL itself needs one GPU per rank
    if "BALER_AMD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BALER_AMD_FORCE_DEVICE"])
    backend = backend or os.environ.get("BALER_AMD_DIST_BACKEND")
    if world > 1 and not is_dist():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        if backend == "nccl":
            torch.cuda.set_device(local)
            td.init_process_group(backend, rank=rank, world_size=world,
                                  device_id=torch.device("cuda", local))
        else:
            if torch.cuda.is_available():
                torch.cuda.set_device(local)
            td.init_process_group(backend, rank=rank, world_size=world)
    return rank, world, local


def allreduce_sum(t):
    """In-place SUM all-reduce on the current stream (RCCL: one ncclAllReduce(ncclSum))."""
    if is_dist() and td.get_world_size() > 1:
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def broadcast(t, src=0):
    """Broadcast a tensor from `src` (initial parameters: every rank constructs its own randomly initialised model)."""
    if is_dist() and td.get_world_size() > 1:
        td.broadcast(t, src=src)
    return t


def barrier():
    if is_dist():
        td.barrier()


def shard_rows(n_rows, rank=None, world=None):
    """Contiguous row range [lo, hi) of rank for collective-free sharding (compress/decompress)."""
    if rank is None:
        rank, world = rank_world()
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
