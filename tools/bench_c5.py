#!/usr/bin/env python3
"""C5 (BASELINE.json configs[4]): 512-column synthetic table, AE(512, 6), fp32, encode-only throughput on one GPU."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import synth
from baler_amd.modules import models
n = int(sys.argv[1]) if len(sys.argv) > 1 else 524288
x = torch.as_tensor(synth.wide_rows(n, 512).astype(np.float32)).cuda()
torch.manual_seed(0)
m = models.CFD_dense_AE(512, 6, mode="fp32").to("cuda:0")
h = m.handle()
z = h.encode(x)
torch.cuda.synchronize()
t0 = time.perf_counter()
K = 5
for _ in range(K):
    z = h.encode(x)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / K
print(f"C5 encode: {n / dt / 1e6:.1f} M rows/s, {dt * 1e3:.2f} ms per {n} rows, {255400 * n / dt / 1e12:.1f} TFLOP/s algorithmic "
      f"({100 * 255400 * n / dt / 157.3e12:.0f}% of fp32 MFMA peak), input stream {2048 * n / dt / 1e12:.2f} TB/s")
