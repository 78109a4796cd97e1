#!/usr/bin/env python3
"""C4 (BASELINE.json configs[3]): CFD 2-D field, CFD_dense_AE(2500, 25), fp32, one GPU: encode and train-step throughput
on the generic layer-wise MFMA path (N frames of 50x50)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import synth
from baler_amd.modules import models
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
x = torch.as_tensor(synth.cfd_field(n).reshape(n, 2500).astype(np.float32)).cuda()
torch.manual_seed(0)
m = models.CFD_dense_AE(2500, 25, mode="fp32").to("cuda:0")
h = m.handle()
grads, mm, vv = torch.zeros_like(m.flat), torch.zeros_like(m.flat), torch.zeros_like(m.flat)
def timeit(fn, k=3):
    fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(k): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
te = timeit(lambda: h.encode(x))
z = h.encode(x)
td = timeit(lambda: h.decode(z))
tt = timeit(lambda: h.fwd_bwd(x, grads))
print(f"C4 N={n}: encode {n / te / 1e6:.2f} M rows/s ({1052500 * n / te / 1e12:.1f} TFLOP/s, {100 * 1052500 * n / te / 157.3e12:.0f}% of fp32 MFMA peak); "
      f"decode {n / td / 1e6:.2f} M rows/s ({100 * 1052500 * n / td / 157.3e12:.0f}%); "
      f"train fwd_bwd {n / tt / 1e6:.2f} M rows/s ({5315000 * n / tt / 1e12:.1f} TFLOP/s, {100 * 5315000 * n / tt / 157.3e12:.0f}%), {tt * 1e3:.1f} ms")
