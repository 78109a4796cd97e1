#!/usr/bin/env python3
"""Optimiser steps of a wide model at the reference's own batch sizes (CFD configs: 60 / 6000 frames, exafel 1 .. 36, hurricane 85):
python tools/bench_wide_small_step.py [F] [Z] [ROWS] [STEPS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
Z = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 64
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 100
os.environ["BALER_AMD_QUIET"] = "1"
dims = orc.ae_dims(F, Z)
h = native.Handle(dims, "fp32")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
m, v = torch.zeros_like(p), torch.zeros_like(p)
x = torch.rand((rows * 8, F), dtype=torch.float32, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): h.train_step(x[(i % 8) * rows:(i % 8 + 1) * rows], p, m, v, i + 1, 1e-3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"AE({F},{Z}) [{h.path}] train_step {rows} rows: {dt * 1e6:.1f} us/step")
