#!/usr/bin/env python3
"""us per bamd_train_step at small batch sizes, 4-row chain (lat4_chain_kernel) vs 16-row chain (BALER_AMD_LAT4_ROWS=0):
   gpurun -- python tools/bench_small_batches.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp32")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
m, v = torch.zeros_like(p), torch.zeros_like(p)
x = torch.rand((1 << 20, 24), dtype=torch.float64, device="cuda")
out = []
for bs in (64, 256, 512, 1024, 2048, 4096, 8192, 12288):
    nb = 400
    best = 1e9
    for rep in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(nb): h.train_step(x[(i * bs) %% (1 << 19):(i * bs) %% (1 << 19) + bs], p, m, v, i + 1, 1e-3)
        torch.cuda.synchronize(); best = min(best, (time.perf_counter() - t0) / nb)
    out.append("%%d:%%.1f" %% (bs, best * 1e6))
print("RES", " ".join(out))
''' % R
for name, env in (("lat4 (<= 1M rows)", "1000000"), ("lat2 only", "0")):
    o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BALER_AMD_LAT4_ROWS=env), capture_output=True, text=True)
    print(f"{name:20s}", [l for l in o.stdout.splitlines() if l.startswith("RES")] or o.stderr[-400:])
