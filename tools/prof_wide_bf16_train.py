#!/usr/bin/env python3
"""Training passes of a BAMD_MODE_BF16 CFD_dense_AE(2500, 25) handle at 32,768 frames, for rocprofv3 (no child processes)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
dims = orc.ae_dims(2500, 25)
h = native.Handle(dims, "bf16")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
x = torch.rand((n, 2500), dtype=torch.float32, device="cuda")
g = torch.zeros_like(p)
for _ in range(6):
    h.fwd_bwd(x, g)
torch.cuda.synchronize()
print("done", float(g[-1]))
