"""Data-parallel plumbing: one process per GPU, ``torch.distributed`` (backend "nccl" = RCCL over xGMI
on the GPU box, "gloo" in CPU tests).  The hot path has exactly one exchange: a SUM all-reduce of the
flat ``[grads | loss]`` buffer per optimiser step (the loss is a row SUM, so the global-batch gradient
is the sum -- not the mean -- of shard gradients).  compress / decompress shard
rows with no data-path collective; off the hot path there are two more exchanges, once per file: a
MIN and a MAX all-reduce of the per-column extrema (2 x C doubles) when the table is row-sharded, and
the rank-ordered gather of the encoded / decoded shards onto rank 0, which writes the artefact.

Global-batch policy (``batch_policy``): ``config.batch_size`` is the GLOBAL batch by default -- a DP run
takes exactly the optimiser steps of the single-process run with the same config, each rank computing
1/N of every batch (strict; what the reference's config means).  With ``config.dp_batch = "per_gpu"`` (or
``BALER_AMD_DP_BATCH=per_gpu``) every GPU takes ``batch_size`` rows per step and the global batch is
N x batch_size: the run equals the single-process reference run with ``batch_size = N x 512`` (fixture g12
for N = 8), and a step keeps one GPU as busy as it is in the single-GPU run.
"""
import os
import sys

import torch
import torch.distributed as td


def is_dist():
    return td.is_available() and td.is_initialized()


def rank_world():
    if is_dist():
        return td.get_rank(), td.get_world_size()
    return 0, 1


def force_pg():
    """BALER_AMD_FORCE_PG=1: build the process group and keep every collective of the data-parallel step even at world size 1
    (one GPU): RCCL's library load, communicator set-up, its stream hand-off around bamd_fwd_bwd / bamd_adam_step and the
    dmabuf IPC mode are then exercised on a single-GPU box exactly as an 8-GPU run exercises them."""
    return os.environ.get("BALER_AMD_FORCE_PG") == "1"


def collectives_on():
    """Does a training step run the data-parallel sequence (bamd_fwd_bwd -> all-reduce -> bamd_adam_step)?  With more than one
    rank always; with ONE rank only when BALER_AMD_FORCE_PG=1 built a group (a sum over one rank is the identity: results are
    bit-identical to the single-process step, tests/test_gpu_dp.py::test_rccl_world1_*)."""
    return is_dist() and (td.get_world_size() > 1 or force_pg())


def init_from_env(backend=None):
    """Initialise the default process group from torchrun's environment (RANK/WORLD_SIZE/MASTER_*).
    Returns (rank, world, local_rank).  No-op for single-process runs unless BALER_AMD_FORCE_PG=1 (then a one-rank group)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    # test hooks: several ranks may share one GPU (BALER_AMD_FORCE_DEVICE) over gloo (BALER_AMD_DIST_BACKEND)
    # so that the multi-rank code path can be exercised on a single-GPU box; RCCL itself needs one GPU per rank
    if "BALER_AMD_FORCE_DEVICE" in os.environ:
        local = int(os.environ["BALER_AMD_FORCE_DEVICE"])
    backend = backend or os.environ.get("BALER_AMD_DIST_BACKEND")
    if (world > 1 or force_pg()) and not is_dist():
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


def lib_comm_wanted():
    """Should the library run the data-parallel step itself (fwd_bwd -> ncclAllReduce -> Adam in one host call)?  Yes on RCCL
    ("nccl" backend, one GPU per rank); BALER_AMD_LIB_COMM=0 keeps the three-call Python sequence.  The gloo test set-ups (several
    ranks sharing one GPU) always keep it: an RCCL communicator needs one device per rank."""
    return (collectives_on() and td.get_backend() == "nccl" and os.environ.get("BALER_AMD_LIB_COMM", "1") != "0")


def attach_comm(handle):
    """Give `handle` its own RCCL communicator over the ranks of the default group: rank 0's ncclGetUniqueId travels through the
    existing process group (a 128-byte broadcast), then every rank joins with ncclCommInitRank inside the library.  Idempotent."""
    from . import native
    if handle.comm_world:
        return handle
    rank, world = rank_world()
    box = [native.comm_unique_id() if rank == 0 else None]
    td.broadcast_object_list(box, src=0)
    handle.comm_init(box[0], rank, world)
    if rank == 0 and os.environ.get("BALER_AMD_QUIET") != "1":
        print(f"[baler_amd] data-parallel step inside the library: RCCL communicator of {world} rank(s) "
              "(bamd_train_epoch_dp: fwd_bwd -> ncclAllReduce -> Adam per batch, one host call per epoch)", file=sys.stderr, flush=True)
    return handle


def allreduce_sum(t):
    """In-place SUM all-reduce on the current stream (RCCL: one ncclAllReduce(ncclSum))."""
    if collectives_on():
        td.all_reduce(t, op=td.ReduceOp.SUM)
    return t


def broadcast(t, src=0):
    """Broadcast a tensor from `src` (initial parameters: every rank constructs its own randomly initialised model)."""
    if collectives_on():
        td.broadcast(t, src=src)
    return t


def all_gather_rows(local, counts):
    """Rank-ordered concatenation of per-rank row blocks ON EVERY RANK (counts[r] = rows of rank r; they differ by at most one in the
    batch split, so the blocks travel padded to the largest).  The sliced-Wasserstein loss sorts the latent codes of ONE global batch."""
    world = len(counts)
    if world == 1 or not is_dist():
        return local
    pad = max(counts)
    send = local.contiguous()
    if local.shape[0] != pad:
        send = torch.zeros((pad,) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = torch.empty((world, pad) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
    td.all_gather_into_tensor(recv.view((world * pad,) + tuple(local.shape[1:])), send) if td.get_backend() == "nccl" \
        else td.all_gather(list(recv.unbind(0)), send)
    return torch.cat([recv[r, :counts[r]] for r in range(world)], dim=0)


def barrier():
    if is_dist():
        td.barrier()


def batch_policy(config=None):
    """-> "global" (config.batch_size is the global batch; default) or "per_gpu" (global = world x batch_size)."""
    v = os.environ.get("BALER_AMD_DP_BATCH") or getattr(config, "dp_batch", None) or "global"
    if v not in ("global", "per_gpu"):
        raise ValueError(f"dp_batch must be 'global' or 'per_gpu', got {v!r}")
    return v


def global_batch(config, world=None):
    """Rows per optimiser step over all ranks."""
    if world is None:
        world = rank_world()[1]
    bs = int(config.batch_size)
    return bs * world if (world > 1 and batch_policy(config) == "per_gpu") else bs


def allreduce_minmax(mm):
    """mm = [min ; max] (2, C) float64 of this rank's rows -> the extrema over all ranks, in place: one MIN and one
    MAX all-reduce of C doubles (exact in any order).  np.min / np.max propagate a NaN (a column with a NaN cell has
    min = max = NaN in data_processing.find_minmax); what a collective's MIN / MAX makes of one is not specified, so
    the NaN columns travel as a third, tiny MAX all-reduce of a 0/1 flag and are poisoned again afterwards."""
    if is_dist() and td.get_world_size() > 1:
        nan = torch.isnan(mm).any(dim=0)
        flag = nan.to(mm.dtype)
        mm[0].masked_fill_(nan, float("inf"))
        mm[1].masked_fill_(nan, float("-inf"))
        td.all_reduce(mm[0], op=td.ReduceOp.MIN)
        td.all_reduce(mm[1], op=td.ReduceOp.MAX)
        td.all_reduce(flag, op=td.ReduceOp.MAX)
        mm[:, flag > 0] = float("nan")
    return mm


def gather_rows(local, n_total, dst=0):
    """Rank-ordered concatenation of contiguous row shards (shard_rows) onto rank `dst`: ONE gather of device tensors
    padded to the largest shard (RCCL: N-1 point-to-point transfers GPU to GPU over xGMI, straight into `dst`'s
    buffer; no pickling, no host round trip per rank).  Returns the (n_total, ...) tensor on `dst`, None elsewhere."""
    if not (is_dist() and td.get_world_size() > 1):
        return local
    rank, world = td.get_rank(), td.get_world_size()
    pad = (n_total + world - 1) // world                     # shard_rows gives base or base + 1 rows: <= ceil(n / world)
    tail = tuple(local.shape[1:])
    send = local.contiguous()
    if local.shape[0] != pad:
        send = torch.zeros((pad,) + tail, dtype=local.dtype, device=local.device)
        send[:local.shape[0]] = local
    recv = parts = None
    if rank == dst:
        recv = torch.empty((world, pad) + tail, dtype=local.dtype, device=local.device)
        parts = list(recv.unbind(0))
    td.gather(send, parts, dst=dst)
    if rank != dst:
        return None
    recv = recv.reshape((world * pad,) + tail)
    if n_total == world * pad:
        return recv
    # ragged total: shards 0..rem-1 have `pad` rows, the others pad - 1.  Compact IN PLACE (rank r's rows move left by r - rem
    # rows, through a copy of one shard: source and destination overlap) instead of concatenating trimmed views, which held a
    # second full copy of the table on rank 0's HBM
    for r in range(world):
        lo, hi = shard_rows(n_total, r, world)
        if lo != r * pad:
            recv[lo:hi].copy_(recv[r * pad:r * pad + (hi - lo)].clone())
    return recv[:n_total]


def shard_rows(n_rows, rank=None, world=None):
    """Contiguous row range [lo, hi) of rank for collective-free sharding (compress/decompress)."""
    if rank is None:
        rank, world = rank_world()
    base, rem = divmod(n_rows, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)
