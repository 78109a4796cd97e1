#!/usr/bin/env python3
"""Mid-batch fp32 optimiser steps of the 24-column model for rocprofv3: 200 bamd_train_step at ROWS rows (argv[1]) behind a clock warm-up."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
from _gpu_warm import warm
rows = int(sys.argv[1])
mode = sys.argv[2] if len(sys.argv) > 2 else "fp32"
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, mode)
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).to(torch.float64 if mode == "fp64" else torch.float32).cuda()
h.load_params(p)
x = torch.from_numpy(orc.normalize(synth.cms_rows(rows * 2))).cuda()
m, v = torch.zeros_like(p), torch.zeros_like(p)
warm(150.0)
for i in range(200):
    k = i % 2
    h.train_step(x[k * rows:(k + 1) * rows], p, m, v, i + 1, 1e-3)
torch.cuda.synchronize()
print("done")
