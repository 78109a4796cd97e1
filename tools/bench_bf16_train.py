#!/usr/bin/env python3
"""Rate of the bf16 training kernels alone (no oracle): python tools/bench_bf16_train.py [rows] [iters]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from baler_amd import native, synth
from baler_amd.modules import models

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 20
raw = torch.as_tensor(synth.cms_rows(rows)).cuda()
xd = native.normalize(raw, native.minmax(raw))
torch.manual_seed(0)
m = models.AE(24, 15, mode="bf16").to("cuda:0")
h = m.handle()
g = torch.zeros_like(m.flat)
for _ in range(3):
    h.fwd_bwd(xd, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(iters):
    h.fwd_bwd(xd, g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / iters
print(f"bf16 fwd_bwd {rows} rows: {dt * 1e3:.3f} ms = {rows / dt / 1e6:.1f} M rows/s = {357000 * rows / dt / 1e12:.1f} TFLOP/s")
