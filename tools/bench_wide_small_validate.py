#!/usr/bin/env python3
"""Validation pass (forward + loss) of a wide model at the reference's batch sizes: python tools/bench_wide_small_validate.py [F] [Z] [ROWS]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
F = int(sys.argv[1]) if len(sys.argv) > 1 else 2500
Z = int(sys.argv[2]) if len(sys.argv) > 2 else 25
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 60
dims = orc.ae_dims(F, Z)
h = native.Handle(dims, "fp32")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
x = torch.rand((rows, F), dtype=torch.float32, device="cuda")
lo = torch.empty(1, dtype=torch.float64, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(100): h.forward_loss(x, want_recon=False, loss_out=lo)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 100
print(f"AE({F},{Z}) validation pass of {rows} rows: {dt * 1e6:.1f} us")
