#!/usr/bin/env python3
"""fp64 kernels for rocprofv3 (no child processes): encode / decode / forward+loss / fwd_bwd at 262,144 rows, 512-row train steps."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
n = 262144
x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
z = h.encode(x)
for _ in range(10):
    z = h.encode(x); y = h.decode(z); h.forward_loss(x, want_recon=False); h.fwd_bwd(x, g)
for i in range(100):
    h.train_step(x[i * 512:(i + 1) * 512], p, m, v, i + 1, 1e-3)
torch.cuda.synchronize()
print("done")
