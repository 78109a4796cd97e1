#!/usr/bin/env python3
"""fp64 bamd_fwd_bwd at 262,144 rows for rocprofv3 (10 launches).  Argument `y`: the wave-owned-tile kernel of the round-6 experiment
(BALER_AMD_DW64Y_BLKS: only in the experiment commits e992b02 / c5eec6f; the shipped library ignores it and runs dw64x_kernel)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
from _gpu_warm import warm
if len(sys.argv) > 1 and sys.argv[1] == "y":
    os.environ["BALER_AMD_DW64Y_BLKS"] = "1024"
n = 262144
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
x = torch.from_numpy(orc.normalize(synth.cms_rows(n))).cuda()
g = torch.zeros_like(p)
warm(100.0)
for _ in range(10):
    h.fwd_bwd(x, g)
torch.cuda.synchronize()
print("done")
