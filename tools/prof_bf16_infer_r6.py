#!/usr/bin/env python3
"""bf16 encode / decode of AE(24, 15) for rocprofv3 with warmed launches: 1M float64 rows (the bench line's `encode_bf16`) and 4M float64 /
float32 rows, a 150-ms clock warm-up (torch matmuls) + 100 launches each; and C5 (512 columns, 262,144 float32 rows) bf16 encode."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native
from baler_amd.modules import models
from oracle import c_oracle as orc
from _gpu_warm import warm
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda())
for n, dt in ((1_000_000, torch.float64), (4_000_000, torch.float64), (4_000_000, torch.float32)):
    x = torch.rand((n, 24), dtype=dt, device="cuda")
    z = h.encode(x); y = h.decode(z)
    for fn in (lambda: h.encode(x, out=z), lambda: h.decode(z, out=y)):
        warm(150.0)
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
    del x, z, y
h.close()
torch.manual_seed(0)
m5 = models.AE(512, 6, mode="bf16").to("cuda:0")
h5 = m5.handle()
x5 = torch.rand((262144, 512), dtype=torch.float32, device="cuda")
z5 = h5.encode(x5, out_dtype=torch.float32)
warm(150.0)
for _ in range(100):
    h5.encode(x5, out=z5)
torch.cuda.synchronize()
print("done")
