#!/usr/bin/env python3
"""fp64 bamd_fwd_bwd at 1M rows against BALER_AMD_F64_CHUNK_ROWS (read at handle creation): do images of a chunk that fits the 256-MB
Infinity Cache (12.7 KB per row: ~16k rows) stay on the die between the chain and the weight-gradient launch?  One process per value."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
torch.manual_seed(1)
n = 1000000
x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
g = torch.zeros_like(p)
for _ in range(3): h.fwd_bwd(x, g)
torch.cuda.synchronize()
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); h.fwd_bwd(x, g); h.fwd_bwd(x, g); e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 2)
ms = sorted(ts)[2]
print("RES %%.3f ms = %%.3f of the fp64 peak, checksum %%.12e" %% (ms, 357000 * n / ms / 1e9 / 78.6, float(g.sum())))
''' % R
for chunk in sys.argv[1:] or ["262144", "131072", "65536", "32768", "16384", "8192"]:
    e = dict(os.environ, BALER_AMD_F64_CHUNK_ROWS=chunk)
    o = subprocess.run([sys.executable, "-c", CHILD], env=e, capture_output=True, text=True)
    print(f"chunk {chunk:>7s} rows:", ([l[4:] for l in o.stdout.splitlines() if l.startswith("RES")] or [o.stderr[-400:]])[0], flush=True)
