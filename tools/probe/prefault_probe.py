import os, sys, time, ctypes, platform
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from concurrent.futures import ThreadPoolExecutor
print(platform.release())
libc = ctypes.CDLL(None, use_errno=True)
libc.madvise.argtypes = [ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int]
n = 10_000_000
a = np.empty((n, 15)); base = a.ctypes.data; lo = (base + 4095) & ~4095
t0 = time.perf_counter(); rc = libc.madvise(lo, 32 << 20, 23); e = ctypes.get_errno(); dt = time.perf_counter() - t0
print(f"madvise(POPULATE_WRITE, 32 MB): rc {rc} errno {e} {dt * 1e3:.2f} ms")
pool = ThreadPoolExecutor(8)
for name, fn in (("madvise", lambda p, l: libc.madvise(p, l, 23)), ("memset", lambda p, l: ctypes.memset(p, 0, l))):
    b = np.empty((n, 15)); base = b.ctypes.data; lo = (base + 4095) & ~4095; hi = (base + b.nbytes) & ~4095
    t0 = time.perf_counter()
    futs = [pool.submit(fn, p, min(32 << 20, hi - p)) for p in range(lo, hi, 32 << 20)]
    [f.result() for f in futs]
    dt = time.perf_counter() - t0
    t0 = time.perf_counter(); b[:] = 1.0; dt2 = time.perf_counter() - t0
    print(f"{name}: 8 threads {b.nbytes / 1e9 / dt:.1f} GB/s; then a full write by one thread: {b.nbytes / 1e9 / dt2:.1f} GB/s")
c = np.empty((n, 15)); t0 = time.perf_counter(); c[:] = 1.0; dt = time.perf_counter() - t0
print(f"full write of a fresh array by one thread: {c.nbytes / 1e9 / dt:.1f} GB/s")
