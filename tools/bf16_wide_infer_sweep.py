#!/usr/bin/env python3
"""bf16 encode / decode of the wide models against the row count: C4 = CFD_dense_AE(2500, 25), C5 = the 512-column model (latent 6).
Back-to-back launches behind a clock warm-up, median of five samples (bench.event_ms's method)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
from _gpu_warm import warm
def ms(fn, reps):
    warm(40.0)
    for _ in range(5): fn()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[2]
for name, F, Z, ns in (("C4 2500-25", 2500, 25, (8192, 32768, 65536, 131072, 262144)), ("C5 512-6", 512, 6, (65536, 262144, 1048576, 2097152))):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "bf16")
    torch.manual_seed(0)
    p = (torch.randn(sum((dims[i] + 1) * dims[i + 1] for i in range(8)) + 1) * 0.05).float().cuda()
    h.load_params(p)
    for n in ns:
        x = torch.rand((n, F), dtype=torch.float32, device="cuda")
        z = torch.empty((n, Z), dtype=torch.float32, device="cuda")
        y = torch.empty((n, F), dtype=torch.float32, device="cuda")
        reps = max(3, min(50, int(2e9 / (n * F * 4))))
        te = ms(lambda: h.encode(x, out=z), reps)
        td = ms(lambda: h.decode(z, out=y), reps)
        gb = n * (F + Z) * 4 / 1e9
        print(f"{name} {n:8d} rows: encode {te * 1e3:8.1f} us = {gb / te:6.0f} GB/s ({gb / te / 8000:.3f} of HBM)   decode {td * 1e3:8.1f} us = {gb / td:6.0f} GB/s ({gb / td / 8000:.3f})", flush=True)
        del x, y, z
    h.close()
