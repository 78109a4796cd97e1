"""Host side of the data pipeline: the .npz table <-> HBM.

The reference loads the whole ``.npz`` with ``np.load`` (helper.py:283-289, 489-492), keeps a second
normalised copy, and moves 512-row batches between host and device one at a time, growing the result
with ``np.concatenate`` (helper.py:583-611, 700-723).  Here:

* ``open_npz_array`` maps a STORED (uncompressed, what ``np.savez`` writes) member of the archive
  straight from the file, so a rank reads only the rows it owns (data-parallel training keeps 1/N of
  the table per GPU, compress / decompress 1/N of the rows); compressed members fall back to
  ``np.load``;
* ``upload_rows`` streams rows (a contiguous range, a block-cyclic shard or an index list) into ONE
  device tensor through two pinned staging buffers: the file read / gather of chunk k+1 runs while
  the DMA engine copies chunk k (a copy stream; the compute stream waits on one event at the end);
* ``download_rows`` is the mirror image for results: D2H of chunk k on the copy stream while the host
  moves chunk k-1 from pinned memory into the final array.  Each chunk can wait for its own
  "producer done" event, so the encode of block k+1 overlaps the download of block k.

PyTorch is used for pinned host memory, streams and events only.
"""
import zipfile

import numpy as np
import torch

CHUNK_BYTES = 64 << 20      # staging buffer size: 2.5 ms of PCIe Gen5 per chunk, launch overheads amortised
import os

# host threads that fill / drain a staging buffer (numpy releases the GIL inside large copies).  Measured on the MI355X box
# (2 x EPYC 9575F): the drain of a 64-MB staging buffer WHILE the DMA engine fills the other one moves 31 GB/s with 4 threads,
# 42 with 8, 49 with 16 (alone: 78 GB/s with 4): the D2H pipeline was bound by these copies, not by PCIe (56 GB/s) and not by
# page faults (tools/probe/d2h_pipe_probe.py)
COPY_THREADS = max(4, min(16, (os.cpu_count() or 8) // 2))
_POOL = None

NP_OF_TORCH = {torch.float32: np.float32, torch.float64: np.float64, torch.float16: np.float16,
               torch.uint8: np.uint8, torch.int32: np.int32, torch.int64: np.int64}


def _pool():
    global _POOL
    if _POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _POOL = ThreadPoolExecutor(max_workers=COPY_THREADS)
    return _POOL


def _parallel(fn, start, stop, min_rows=4096):
    """fn(a, b) over [start, stop) cut into COPY_THREADS contiguous pieces (inline when the range is small)."""
    n = stop - start
    if COPY_THREADS <= 1 or n < 2 * min_rows:
        fn(start, stop)
        return
    step = -(-n // COPY_THREADS)
    futs = [_pool().submit(fn, a, min(a + step, stop)) for a in range(start, stop, step)]
    for f in futs:
        f.result()


_FAULT_POOL = None
PREFAULT_THREADS = 8


PREFAULT_SPAN = 32 << 20


def prefault(arr):
    """Map the pages of a freshly allocated host array on background threads, AHEAD of the copies that fill it:
    ctypes.memset per span (the call releases the GIL; a numpy strided store per page did not and serialised everything;
    madvise(MADV_POPULATE_WRITE) maps 4-KiB pages at 27 GB/s and leaves the array 3x slower to write than pages that
    came in through transparent huge pages -- memset: 113 GB/s on 8 threads, tools/probe/prefault_probe.py).
    Measured on the MI355X box: download_rows into a fresh 1.2 GB array 15 GB/s (the drain copies page-fault it in
    while the DMA engine competes for the memory system) against 49 GB/s into a mapped one.
    Returns [(start_byte, end_byte, future)] in address order for wait_prefault(); [] when the array is small.
    Every job holds a reference to ``arr``: the memory cannot be freed under a running memset, whatever the caller does."""
    global _FAULT_POOL
    nbytes = arr.nbytes
    if nbytes < (64 << 20):
        return []
    import ctypes
    base = arr.ctypes.data
    if _FAULT_POOL is None:
        from concurrent.futures import ThreadPoolExecutor
        _FAULT_POOL = ThreadPoolExecutor(max_workers=PREFAULT_THREADS)

    def job(keep, addr, count):
        ctypes.memset(addr, 0, count)
        return keep is not None

    return [(a, min(a + PREFAULT_SPAN, nbytes), _FAULT_POOL.submit(job, arr, base + a, min(PREFAULT_SPAN, nbytes - a)))
            for a in range(0, nbytes, PREFAULT_SPAN)]


def wait_prefault(futs, stop_byte=None):
    """Block until the spans that START below ``stop_byte`` are mapped (``None``: all of them)."""
    while futs and (stop_byte is None or futs[0][0] < stop_byte):
        futs.pop(0)[2].result()


def open_npz_array(path, key="data"):
    """Array ``key`` of an .npz archive WITHOUT reading it: a read-only ``np.memmap`` when the member is
    stored uncompressed (``np.savez``), else the loaded array (``np.savez_compressed``)."""
    with zipfile.ZipFile(path) as zf:
        info = zf.getinfo(key + ".npy")
        if info.compress_type != zipfile.ZIP_STORED:
            with zf.open(info) as f:
                return np.lib.format.read_array(f, allow_pickle=False)
        with open(path, "rb") as raw:
            raw.seek(info.header_offset)
            local = raw.read(30)                              # local file header: name / extra lengths at 26, 28
            if local[:4] != b"PK\x03\x04":
                raise ValueError(f"{path}: bad local header for {key}.npy")
            name_len = int.from_bytes(local[26:28], "little")
            extra_len = int.from_bytes(local[28:30], "little")
            raw.seek(info.header_offset + 30 + name_len + extra_len)
            version = np.lib.format.read_magic(raw)
            if version == (1, 0):
                shape, fortran, dtype = np.lib.format.read_array_header_1_0(raw)
            else:
                shape, fortran, dtype = np.lib.format.read_array_header_2_0(raw)
            offset = raw.tell()
    if fortran or dtype.hasobject:
        return np.load(path, allow_pickle=False)[key]
    if int(np.prod(shape)) == 0:
        return np.zeros(shape, dtype=dtype)
    return np.memmap(path, dtype=dtype, mode="r", offset=offset, shape=tuple(shape))


class RowPlan:
    """Which rows of a table one rank keeps, in the order it keeps them.

    kind "range":  rows [lo, hi)
    kind "cyclic": for every global batch b of `batch` rows, rows [b*batch + a, b*batch + e) with (a, e) the rank's
                   slice of a full batch; the last, partial batch contributes its own slice (training.rank_slice)
    kind "index":  an explicit int64 row index (a train/test split is a permutation)
    ``local_spans`` (cyclic / index-with-batches): [(lo, hi)] of every global batch inside the rank's local tensor."""

    def __init__(self, kind, n_rows, **kw):
        self.kind, self.n_rows = kind, int(n_rows)
        self.__dict__.update(kw)

    @staticmethod
    def whole(n_rows):
        return RowPlan("range", n_rows, lo=0, hi=int(n_rows), count=int(n_rows))

    @staticmethod
    def contiguous(n_rows, rank, world):
        base, rem = divmod(int(n_rows), world)
        lo = rank * base + min(rank, rem)
        hi = lo + base + (1 if rank < rem else 0)
        return RowPlan("range", n_rows, lo=lo, hi=hi, count=hi - lo)

    @staticmethod
    def slice_of(lo, hi, rank, world):
        """Contiguous slice of global batch [lo, hi) owned by `rank` (sizes differ by at most one row)."""
        base, rem = divmod(hi - lo, world)
        a = lo + rank * base + min(rank, rem)
        return a, a + base + (1 if rank < rem else 0)

    @staticmethod
    def cyclic(n_rows, batch, rank, world, index=None):
        """Block-cyclic shard: rank r keeps its slice of EVERY global batch of `batch` rows.  With
        ``index`` (a permutation / subset) the batches are cut from index order."""
        n = int(n_rows if index is None else len(index))
        spans, pieces, off = [], [], 0
        for lo in range(0, n, batch):
            a, e = RowPlan.slice_of(lo, min(lo + batch, n), rank, world)
            spans.append((off, off + e - a))
            pieces.append((a, e))
            off += e - a
        plan = RowPlan("cyclic" if index is None else "index", n_rows, count=off, batch=int(batch), n_global=n,
                       local_spans=spans, pieces=pieces)
        if index is not None:
            idx = np.asarray(index, dtype=np.int64)
            plan.index = np.concatenate([idx[a:e] for a, e in pieces]) if pieces else np.zeros(0, np.int64)
        return plan

    def gather_into(self, src, dst, start, stop):
        """dst[:stop-start] = local rows [start, stop) of this plan taken from src (host arrays, same trailing shape)."""
        m = stop - start
        if self.kind == "range":
            np.copyto(dst[:m], src[self.lo + start:self.lo + stop], casting="unsafe")
            return
        if self.kind == "index":
            np.take(src, self.index[start:stop], axis=0, out=dst[:m]) if src.dtype == dst.dtype else \
                np.copyto(dst[:m], src[self.index[start:stop]], casting="unsafe")
            return
        # cyclic: full batches share one (a, e) offset pair -> ONE strided copy per chunk; then the ragged ends
        a0, e0 = self.pieces[0]
        w = e0 - a0
        nfull = self.n_global // self.batch if w else 0     # batches whose slice is the full-batch slice
        pos = start
        while pos < stop:
            if w and pos < nfull * w:
                b0, r0 = divmod(pos, w)
                if r0 == 0 and stop - pos >= w:
                    nb = min((stop - pos) // w, nfull - b0)
                    view = src[b0 * self.batch:(b0 + nb) * self.batch].reshape((nb, self.batch) + src.shape[1:])
                    np.copyto(dst[pos - start:pos - start + nb * w].reshape((nb, w) + src.shape[1:]),
                              view[:, a0:e0], casting="unsafe")
                    pos += nb * w
                    continue
                take = min(w - r0, stop - pos)
                g = b0 * self.batch + a0 + r0
                np.copyto(dst[pos - start:pos - start + take], src[g:g + take], casting="unsafe")
                pos += take
                continue
            # the partial last batch
            (ls, le), (a, e) = self.local_spans[-1], self.pieces[-1]
            take = min(le - pos, stop - pos)
            g = a + (pos - ls)
            np.copyto(dst[pos - start:pos - start + take], src[g:g + take], casting="unsafe")
            pos += take


_PINNED = {}


def _staging(slot, rows, tail, dtype):
    """Pinned staging buffer `slot` (0/1 upload, 2/3 download) viewed as (rows,) + tail of dtype; the page-locked
    allocation is cached (pinning 64 MB costs ~10 ms) and only ever grows."""
    need = rows * (int(np.prod(tail)) if tail else 1) * torch.empty((), dtype=dtype).element_size()
    buf = _PINNED.get(slot)
    if buf is None or buf.numel() < need:
        buf = torch.empty(need, dtype=torch.uint8).pin_memory()
        _PINNED[slot] = buf
    return buf[:need].view(dtype).view((rows,) + tuple(tail))


def _device_dtype(np_dtype):
    """float32 / float64 tables keep their dtype; anything else is converted to float64 on the host chunk
    (the reference does the same implicitly when it builds float64 tensors, training.py:230)."""
    if np_dtype == np.float32:
        return torch.float32, np.float32
    return torch.float64, np.float64


def upload_rows(src, plan=None, device=None, chunk_bytes=None):
    """Rows of host array ``src`` selected by ``plan`` (default: all) -> ONE contiguous device tensor of shape
    (plan.count,) + src.shape[1:], float32 or float64.  Double-buffered: pinned staging x 2 + a copy stream."""
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device())
    device = torch.device(device)
    plan = plan or RowPlan.whole(src.shape[0])
    t_dtype, h_dtype = _device_dtype(src.dtype)
    tail = tuple(src.shape[1:])
    row_elems = int(np.prod(tail)) if tail else 1
    out = torch.empty((plan.count,) + tail, dtype=t_dtype, device=device)
    if plan.count == 0:
        return out
    row_bytes = row_elems * np.dtype(h_dtype).itemsize
    chunk_rows = max(1, int((chunk_bytes or CHUNK_BYTES) // row_bytes))
    chunk_rows = min(chunk_rows, plan.count)
    if device.type != "cuda":                     # CPU tensors (host-logic tests): one gather, no staging
        plan.gather_into(src, out.numpy(), 0, plan.count)
        return out
    stage = [_staging(i, chunk_rows, tail, t_dtype) for i in range(2)]
    free = [None, None]                           # event: DMA out of the staging buffer finished
    copy_stream = torch.cuda.Stream(device=device)
    with torch.cuda.device(device):
        for k, start in enumerate(range(0, plan.count, chunk_rows)):
            stop = min(start + chunk_rows, plan.count)
            buf = stage[k & 1]
            if free[k & 1] is not None:
                free[k & 1].synchronize()
            host = buf.numpy()
            # file read / gather by several host threads; overlaps the previous chunk's DMA
            _parallel(lambda a, b: plan.gather_into(src, host[a - start:], a, b), start, stop)
            with torch.cuda.stream(copy_stream):
                out[start:stop].copy_(buf[:stop - start], non_blocking=True)
                ev = torch.cuda.Event()
                ev.record(copy_stream)
            free[k & 1] = ev
        torch.cuda.current_stream(device).wait_stream(copy_stream)   # consumers on the compute stream see the rows
    out.record_stream(copy_stream)
    for ev in free:                               # the cached staging buffers may be refilled by the next call
        if ev is not None:
            ev.synchronize()
    return out


def _drain(out, stage, pending, faults=None):
    """Move a landed staging buffer into the result array (several host threads)."""
    b, s0, s1, ev = pending
    if faults:
        wait_prefault(faults, s1 * (out.nbytes // max(1, out.shape[0])))
    ev.synchronize()
    host = stage[b].numpy()
    _parallel(lambda a, e: np.copyto(out[a:e], host[a - s0:e - s0]), s0, s1)


def download_rows(dev, out=None, ready=None, block_rows=None, chunk_bytes=None):
    """Device tensor ``dev`` (n, ...) -> host ndarray (``out`` or a new array).  D2H of chunk k runs on a copy stream
    while the host moves chunk k-1 out of pinned memory.  ``ready``: optional list of (row_stop, event) in row
    order -- rows below ``row_stop`` are complete once ``event`` has fired (lets a producer kernel of block k+1 run
    while block k is downloaded); without it the copy stream waits for the current stream once."""
    n = dev.shape[0]
    tail = tuple(dev.shape[1:])
    np_dtype = NP_OF_TORCH[dev.dtype]
    faults = None
    if out is None:
        out = np.empty((n,) + tail, dtype=np_dtype)
        if dev.is_cuda:
            faults = prefault(out)      # pages mapped ahead of the drain copies, in the background
    if n == 0:
        return out
    if not dev.is_cuda:
        np.copyto(out, dev.numpy())
        return out
    row_bytes = (int(np.prod(tail)) if tail else 1) * dev.element_size()
    chunk_rows = min(n, max(1, int((chunk_bytes or CHUNK_BYTES) // row_bytes)))
    stage = [_staging(2 + i, chunk_rows, tail, dev.dtype) for i in range(2)]
    copy_stream = torch.cuda.Stream(device=dev.device)
    ready = list(ready or [])
    if not ready:
        copy_stream.wait_stream(torch.cuda.current_stream(dev.device))
    pending = None                               # (buffer index, start, stop, event)
    ri = 0
    try:
        with torch.cuda.device(dev.device):
            for k, start in enumerate(range(0, n, chunk_rows)):
                stop = min(start + chunk_rows, n)
                while ri < len(ready) and (ri == 0 or ready[ri - 1][0] < stop):
                    copy_stream.wait_event(ready[ri][1])      # every producer event up to the one covering `stop`
                    ri += 1
                with torch.cuda.stream(copy_stream):
                    stage[k & 1][:stop - start].copy_(dev[start:stop], non_blocking=True)
                    ev = torch.cuda.Event()
                    ev.record(copy_stream)
                if pending is not None:
                    _drain(out, stage, pending, faults)
                pending = (k & 1, start, stop, ev)
            _drain(out, stage, pending, faults)
    finally:
        # no memset may outlive this call: a HIP error or KeyboardInterrupt above would otherwise leave up to
        # PREFAULT_THREADS jobs zeroing spans of an array the caller has already had copied into (or dropped)
        if faults:
            wait_prefault(faults)
    dev.record_stream(copy_stream)
    return out
