"""Clock / power warm-up for profiling scripts: ~`ms` milliseconds of fp32 torch matmuls (NOT our kernels, so rocprofv3's per-kernel averages of
the measured launches are not diluted by ramp-up launches).  The first 20-30 ms of any kernel mix run 10-13 % slow on this chip;
bench.py warms every side measurement with 25-40 ms of the same call -- a 36-launch profile of a 0.3-ms kernel sat entirely inside that ramp."""
import time
import torch


def warm(ms=200.0):
    a = torch.randn((4096, 4096), device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(10):
            a = (a @ a) * 1e-3
        torch.cuda.synchronize()
