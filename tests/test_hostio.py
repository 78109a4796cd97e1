"""CPU tests of the host data pipeline (baler_amd/hostio.py) and of the data-parallel row sharding built on it:
which rows a rank reads from the archive, keeps resident and trains on."""
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

from baler_amd import dist as bdist
from baler_amd import hostio
from baler_amd.modules import training

from conftest import free_port

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_open_npz_array_maps_stored_members(tmp_path):
    rng = np.random.default_rng(0)
    a = rng.normal(size=(1000, 24))
    names = np.array([f"c{i}" for i in range(24)])
    np.savez(tmp_path / "s.npz", data=a, names=names)
    np.savez_compressed(tmp_path / "c.npz", names=names, data=a.astype(np.float32))
    m = hostio.open_npz_array(str(tmp_path / "s.npz"), "data")
    assert isinstance(m, np.memmap) and m.shape == a.shape and m.dtype == a.dtype
    assert np.array_equal(np.asarray(m), a)
    assert np.array_equal(np.asarray(m[137:401]), a[137:401])          # a rank reads a row range only
    c = hostio.open_npz_array(str(tmp_path / "c.npz"), "data")
    assert not isinstance(c, np.memmap) and np.array_equal(c, a.astype(np.float32))
    # 3-D tables and the block view of convert_to_blocks
    f = rng.normal(size=(12, 8, 6))
    np.savez(tmp_path / "f.npz", data=f, names=np.array(["x"]))
    m3 = hostio.open_npz_array(str(tmp_path / "f.npz"))
    assert np.array_equal(np.asarray(m3.reshape(-1, 4, 3)), f.reshape(-1, 4, 3))


def _brute(n, batch, rank, world, index=None):
    idx = np.arange(n) if index is None else np.asarray(index)
    rows, spans, off = [], [], 0
    for lo in range(0, len(idx), batch):
        a, e = training._rank_slice(lo, min(lo + batch, len(idx)), rank, world)
        rows.append(idx[a:e])
        spans.append((off, off + e - a))
        off += e - a
    return np.concatenate(rows) if rows else np.zeros(0, np.int64), spans


@pytest.mark.parametrize("n,batch,world", [(3000, 512, 2), (10_000, 512, 8), (1000, 100, 3), (517, 512, 8), (64, 512, 8),
                                           (4096, 4096, 8), (7, 4, 8)])
def test_cyclic_plan_matches_per_batch_slicing(n, batch, world):
    src = np.arange(n * 3, dtype=np.float64).reshape(n, 3)
    seen = []
    for rank in range(world):
        plan = hostio.RowPlan.cyclic(n, batch, rank, world)
        want_rows, want_spans = _brute(n, batch, rank, world)
        assert plan.count == len(want_rows) and plan.local_spans == want_spans
        for chunk in (1 << 30, 5 * 24, 7 * 24, 24):               # one chunk; chunks that cut slices at odd places
            got = hostio.upload_rows(src, plan, "cpu", chunk_bytes=chunk).numpy()
            assert np.array_equal(got, src[want_rows]), (rank, chunk)
        seen.append(want_rows)
    allrows = np.concatenate(seen)
    assert len(allrows) == n and np.array_equal(np.sort(allrows), np.arange(n))   # the shards tile the table


def test_index_and_range_plans():
    n = 1234
    src = np.random.default_rng(1).normal(size=(n, 5)).astype(np.float32)
    perm = np.random.RandomState(1).permutation(n)[: 1000]
    for rank in range(3):
        plan = hostio.RowPlan.cyclic(n, 128, rank, 3, index=perm)
        want_rows, want_spans = _brute(n, 128, rank, 3, index=perm)
        assert plan.local_spans == want_spans
        got = hostio.upload_rows(src, plan, "cpu", chunk_bytes=37 * 20)
        assert got.dtype == torch.float32 and np.array_equal(got.numpy(), src[want_rows])
        lo, hi = bdist.shard_rows(n, rank, 3)
        got = hostio.upload_rows(src, hostio.RowPlan.contiguous(n, rank, 3), "cpu", chunk_bytes=100)
        assert np.array_equal(got.numpy(), src[lo:hi])
    ints = np.arange(40, dtype=np.int32).reshape(10, 4)          # non-float tables become float64, like torch.tensor(data)
    got = hostio.upload_rows(ints, None, "cpu")
    assert got.dtype == torch.float64 and np.array_equal(got.numpy(), ints.astype(np.float64))
    back = hostio.download_rows(got)
    assert back.dtype == np.float64 and np.array_equal(back, ints)


def test_sharded_rows_and_local_batches():
    n, bs, world = 3000, 512, 2
    data = torch.arange(n * 2, dtype=torch.float64).reshape(n, 2)
    for rank in range(world):
        plan = hostio.RowPlan.cyclic(n, bs, rank, world)
        local = hostio.upload_rows(data.numpy(), plan, "cpu")
        sh = training.ShardedRows(local, plan.n_global, plan.local_spans, bs, rank, world)
        assert sh.shape == (n, 2)
        rows, spans = training._local_batches(sh, bs, rank, world)
        rep_rows, rep_spans = training._local_batches(data, bs, rank, world)     # replicated residency: same batches
        assert len(spans) == len(rep_spans) == 6
        for (a, b), (ra, rb) in zip(spans, rep_spans):
            assert torch.equal(rows[a:b], rep_rows[ra:rb])
        with pytest.raises(ValueError):
            training._local_batches(sh, 256, rank, world)


def test_batch_policy(monkeypatch):
    class C:
        batch_size = 512
    monkeypatch.delenv("BALER_AMD_DP_BATCH", raising=False)
    assert bdist.global_batch(C, 8) == 512 and bdist.global_batch(C, 1) == 512
    C.dp_batch = "per_gpu"
    assert bdist.global_batch(C, 8) == 4096 and bdist.global_batch(C, 1) == 512
    monkeypatch.setenv("BALER_AMD_DP_BATCH", "global")
    assert bdist.global_batch(C, 8) == 512
    monkeypatch.setenv("BALER_AMD_DP_BATCH", "nope")
    with pytest.raises(ValueError):
        bdist.global_batch(C, 8)


_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
from baler_amd import dist as bdist, hostio
rank, world, _ = bdist.init_from_env("gloo")
n = 1001
table = np.random.default_rng(3).normal(size=(n, 6))
lo, hi = bdist.shard_rows(n, rank, world)
# gather_rows: rank-ordered shards -> rank 0 only
full = bdist.gather_rows(torch.from_numpy(table[lo:hi].copy()), n, dst=0)
if rank == 0:
    assert np.array_equal(full.numpy(), table)
else:
    assert full is None
# column extrema of a sharded table == those of the whole table, bit for bit
mine = torch.from_numpy(table[lo:hi])
mm = torch.stack([mine.min(0).values, mine.max(0).values])
bdist.allreduce_minmax(mm)
assert np.array_equal(mm[0].numpy(), table.min(0)) and np.array_equal(mm[1].numpy(), table.max(0))
# a NaN cell on ONE rank poisons its column on every rank (np.min / np.max propagate it), the other columns stay exact
t2 = table.copy(); t2[700, 4] = np.nan
mine = torch.from_numpy(t2[lo:hi])
has = torch.isnan(mine).any(0)
mm = torch.stack([mine.min(0).values, mine.max(0).values]); mm[:, has] = float("nan")
bdist.allreduce_minmax(mm)
ok = [c for c in range(6) if c != 4]
assert torch.isnan(mm[:, 4]).all() and np.array_equal(mm[0].numpy()[ok], table.min(0)[ok]) and np.array_equal(mm[1].numpy()[ok], table.max(0)[ok])
# a rank keeps only its slice of every global batch, and the slices tile the table
plan = hostio.RowPlan.cyclic(n, 128, rank, world)
local = hostio.upload_rows(table, plan, "cpu")
assert local.shape[0] == plan.count and abs(plan.count - n / world) <= len(plan.local_spans)
cnt = torch.tensor([float(local.shape[0])]); bdist.allreduce_sum(cnt)
s = local.sum(0); bdist.allreduce_sum(s)
assert int(cnt.item()) == n and np.allclose(s.numpy(), table.sum(0), rtol=1e-12)
if rank == 0:
    print("HOSTIO-DP-OK")
bdist.barrier()
'''


def test_sharding_collectives_world3_gloo(tmp_path):
    script = tmp_path / "w.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=REPO, OMP_NUM_THREADS="1", BALER_AMD_FORCE_DEVICE="0")      # (on a one-GPU box all three ranks share device 0)
    out = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=3", "--master-addr",
                          "127.0.0.1", "--master-port", str(free_port()), str(script)],
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "HOSTIO-DP-OK" in out.stdout


def test_bench_spawns_its_own_ranks(tmp_path):
    """`python bench.py --gpus 2` outside torchrun: the parent (which must not touch the GPU) starts two ranks as a child
    process and relays the child's exit code.  Here there is no GPU, so the ranks fail loudly -- what is checked is that
    two ranks were started (WORLD_SIZE=2 in the children) and that the parent returns the failure."""
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_gpu_dp.py")
    env = dict(os.environ, BALER_AMD_DIST_BACKEND="gloo", BALER_AMD_FORCE_DEVICE="0")
    env.pop("WORLD_SIZE", None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--no-cpu-baseline", "--no-extras"],
                       env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "no MI355X" in r.stderr and "nproc" not in r.stdout
    assert r.stderr.count("NativeError") >= 2 or "local_rank: 1" in r.stderr or "rank: 1" in r.stderr


def test_prefault_jobs_hold_the_array_and_wait_by_span_start():
    """hostio.prefault: every PENDING or RUNNING background memset keeps its array alive (a finished job has dropped its
    reference), and wait_prefault(stop) waits for every span that STARTS below `stop` (a span overlapping the rows about to be
    drained must be mapped before they are copied in).  Deterministic: the pool is blocked behind gate jobs while the array's
    liveness is checked, so nothing depends on how fast the memsets run."""
    import gc
    import threading
    import weakref
    from concurrent.futures import ThreadPoolExecutor
    if hostio._FAULT_POOL is None:
        hostio._FAULT_POOL = ThreadPoolExecutor(max_workers=hostio.PREFAULT_THREADS)
    gate = threading.Event()
    gates = [hostio._FAULT_POOL.submit(gate.wait) for _ in range(hostio.PREFAULT_THREADS)]      # every worker is parked
    try:
        n = (3 * hostio.PREFAULT_SPAN + 12345) // 8
        arr = np.empty(n, dtype=np.float64)
        ref = weakref.ref(arr)
        futs = hostio.prefault(arr)
        assert [(a, e) for a, e, _ in futs] == [(a, min(a + hostio.PREFAULT_SPAN, arr.nbytes))
                                                for a in range(0, arr.nbytes, hostio.PREFAULT_SPAN)]
        del arr
        gc.collect()
        assert ref() is not None and not any(f.done() for _, _, f in futs)      # pending jobs hold the array
    finally:
        gate.set()
    for g in gates:
        g.result()
    hostio.wait_prefault(futs, hostio.PREFAULT_SPAN + 1)
    assert len(futs) == 2                         # spans starting at 0 and at PREFAULT_SPAN are done and popped
    hostio.wait_prefault(futs)
    assert futs == []
    gc.collect()
    assert ref() is None                          # every job has finished and dropped its reference: the array is gone
    assert hostio.prefault(np.empty(1024)) == []  # small arrays are not worth a thread hop
