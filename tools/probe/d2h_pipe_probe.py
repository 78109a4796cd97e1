import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from baler_amd import hostio
n = 10_000_000
dev = torch.rand((n, 15), dtype=torch.float64, device="cuda")
out = np.zeros((n, 15))
rows = (64 << 20) // 120
stage = [hostio._staging(2 + i, rows, (15,), torch.float64) for i in range(2)]
side = torch.cuda.Stream()
side.wait_stream(torch.cuda.current_stream())
torch.cuda.synchronize()
T = {"issue": 0.0, "wait": 0.0, "copy": 0.0}
t00 = time.perf_counter()
pending = None
for k, s in enumerate(range(0, n, rows)):
    e = min(s + rows, n)
    t0 = time.perf_counter()
    with torch.cuda.stream(side):
        stage[k & 1][:e - s].copy_(dev[s:e], non_blocking=True)
        ev = torch.cuda.Event(); ev.record(side)
    t1 = time.perf_counter(); T["issue"] += t1 - t0
    if pending is not None:
        b, s0, s1, pev = pending
        pev.synchronize(); t2 = time.perf_counter(); T["wait"] += t2 - t1
        host = stage[b].numpy()
        hostio._parallel(lambda a, c: np.copyto(out[a:c], host[a - s0:c - s0]), s0, s1)
        T["copy"] += time.perf_counter() - t2
    pending = (k & 1, s, e, ev)
b, s0, s1, pev = pending
pev.synchronize(); host = stage[b].numpy(); np.copyto(out[s0:s1], host[:s1 - s0])
dt = time.perf_counter() - t00
print(f"total {dt * 1e3:.1f} ms = {n * 120 / 1e9 / dt:.1f} GB/s;", {k: f"{v * 1e3:.1f} ms" for k, v in T.items()})
for thr in (8, 16, 32):
    hostio.COPY_THREADS = thr; hostio._POOL = None
    torch.cuda.synchronize(); t0 = time.perf_counter(); z = hostio.download_rows(dev, out=out); dt = time.perf_counter() - t0
    t0 = time.perf_counter(); z = hostio.download_rows(dev, out=out); dt = time.perf_counter() - t0
    print(f"download_rows, {thr} copy threads: {n * 120 / 1e9 / dt:.1f} GB/s")
t0 = time.perf_counter(); p = torch.empty((n, 15), dtype=torch.float64).pin_memory(); dt = time.perf_counter() - t0
print(f"torch.empty + pin_memory 1.2 GB: {dt * 1e3:.0f} ms")
t0 = time.perf_counter(); p2 = torch.empty((n, 15), dtype=torch.float64, pin_memory=True); dt = time.perf_counter() - t0
print(f"torch.empty(pin_memory=True) 1.2 GB: {dt * 1e3:.0f} ms")
torch.cuda.synchronize(); t0 = time.perf_counter(); p2.copy_(dev, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"one DMA into the pinned result: {n * 120 / 1e9 / dt:.1f} GB/s")
a = np.empty((n, 15)); 
rt = torch.cuda.cudart()
t0 = time.perf_counter(); rc = rt.cudaHostRegister(a.ctypes.data, a.nbytes, 0); dt = time.perf_counter() - t0
print(f"hipHostRegister of a fresh 1.2 GB numpy array: rc {rc} {dt * 1e3:.0f} ms")
ta = torch.from_numpy(a)
torch.cuda.synchronize(); t0 = time.perf_counter(); ta.copy_(dev, non_blocking=True); torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"one DMA into the registered array: {n * 120 / 1e9 / dt:.1f} GB/s, is_pinned {ta.is_pinned()}")
t0 = time.perf_counter(); rt.cudaHostUnregister(a.ctypes.data); print(f"unregister {(time.perf_counter() - t0) * 1e3:.0f} ms")
