#!/usr/bin/env python3
"""The run-time-width wide class for rocprofv3: CFD_dense_AE(900, 9), 131,072 rows: encode / decode / fwd_bwd, a 150-ms clock warm-up (torch matmuls) + 100 launches each."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
from _gpu_warm import warm
dims = orc.ae_dims(900, 9)
h = native.Handle(dims, "fp32")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
assert h.path == "fused"
x = torch.rand((131072, 900), dtype=torch.float32, device="cuda")
g = torch.zeros_like(p)
z = h.encode(x)
y = torch.empty_like(x)
for fn in (lambda: h.encode(x, out=z), lambda: h.decode(z, out=y), lambda: h.fwd_bwd(x, g)):
    warm(150.0)
    for _ in range(100):
        fn()
    torch.cuda.synchronize()
print("done")
