"""Where does the D2H pipeline's time go?  gpurun -- python tools/probe/d2h_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from baler_amd import hostio
n = 10_000_000
dev = torch.rand((n, 15), dtype=torch.float64, device="cuda")
gb = dev.numel() * 8 / 1e9
pin = torch.empty((1 << 26) // 8, dtype=torch.float64).pin_memory()
torch.cuda.synchronize()
# 1. pure DMA into one pinned 64-MB buffer, back to back
flat = dev.view(-1)
t0 = time.perf_counter()
for s in range(0, flat.numel(), pin.numel()):
    e = min(s + pin.numel(), flat.numel())
    pin[:e - s].copy_(flat[s:e], non_blocking=True)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"DMA device -> pinned (64 MB pieces): {gb / dt:.1f} GB/s")
# 3. memcpy pinned -> faulted array
a = np.zeros((n, 15))
src = pin.numpy()
t0 = time.perf_counter()
av = a.reshape(-1)
for s in range(0, av.size, src.size):
    e = min(s + src.size, av.size)
    hostio._parallel(lambda x, y: np.copyto(av[x:y], src[x - s:y - s]), s, e)
dt = time.perf_counter() - t0
print(f"copy pinned -> mapped array, {hostio.COPY_THREADS} threads: {gb / dt:.1f} GB/s")
c = np.empty((n, 15)); cv = c.reshape(-1)
t0 = time.perf_counter()
for s in range(0, cv.size, src.size):
    e = min(s + src.size, cv.size)
    hostio._parallel(lambda x, y: np.copyto(cv[x:y], src[x - s:y - s]), s, e)
dt = time.perf_counter() - t0
print(f"copy pinned -> FRESH array, {hostio.COPY_THREADS} threads: {gb / dt:.1f} GB/s")
# 4. the pipeline
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); z = hostio.download_rows(dev); dt = time.perf_counter() - t0
    print(f"download_rows (fresh array): {gb / dt:.1f} GB/s")
torch.cuda.synchronize(); t0 = time.perf_counter(); z = hostio.download_rows(dev, out=a); dt = time.perf_counter() - t0
print(f"download_rows (mapped array): {gb / dt:.1f} GB/s")
print("cpus", os.cpu_count(), open("/sys/kernel/mm/transparent_hugepage/enabled").read().strip())
