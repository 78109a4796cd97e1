#!/usr/bin/env python3
"""N bamd_train_step calls at one batch size (for rocprofv3): python tools/bench_one_batch.py ROWS [STEPS] [MODE]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
bs = int(sys.argv[1]); steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200; mode = sys.argv[3] if len(sys.argv) > 3 else "fp32"
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, mode)
dt_ = torch.float64 if mode == "fp64" else torch.float32
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).to(dt_).cuda()
h.load_params(p)
m, v = torch.zeros_like(p), torch.zeros_like(p)
x = torch.rand((max(bs * 8, 1 << 16), 24), dtype=torch.float64, device="cuda")
for rep in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for i in range(steps): h.train_step(x[(i % 8) * bs:(i % 8) * bs + bs], p, m, v, i + 1, 1e-3)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / steps
print(f"{mode} train_step {bs} rows: {dt * 1e6:.1f} us/step")
