#!/usr/bin/env python3
"""The reference's own regime for rocprofv3 (no child processes): 300 bamd_train_step of 512 float64 rows, BAMD_MODE_F64
(chain64q_kernel + dw64_kernel<adam>)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
x = torch.from_numpy(orc.normalize(synth.cms_rows(512 * 40))).cuda()
m, v = torch.zeros_like(p), torch.zeros_like(p)
for i in range(300):
    k = i % 40
    h.train_step(x[k * 512:(k + 1) * 512], p, m, v, i + 1, 1e-3)
torch.cuda.synchronize()
print("done")
