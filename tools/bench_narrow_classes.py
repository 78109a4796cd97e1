#!/usr/bin/env python3
"""Throughput of narrow tables on the class instantiation (run-time widths) against the exact 24-column instantiation:
encode / decode / fwd_bwd at 1M rows.  python tools/bench_narrow_classes.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
def ms(fn, reps=5):
    for _ in range(10): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
n = 1_000_000
for F, Z in ((24, 15), (25, 10), (31, 15), (16, 4), (33, 8), (47, 15), (24, 16), (47, 31), (48, 12), (63, 31), (64, 16), (79, 31), (80, 16)):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "fp32")
    p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    x = torch.rand((n, F), dtype=torch.float64, device="cuda")
    z = h.encode(x)
    g = torch.zeros_like(p)
    te, td, tt = ms(lambda: h.encode(x)), ms(lambda: h.decode(z)), ms(lambda: h.fwd_bwd(x, g), 3)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    st = {"t": 0}
    def steps():
        for i in range(100):
            st["t"] += 1
            h.train_step(x[i * 512:(i + 1) * 512], p, m, v, st["t"], 1e-3)
    t5 = ms(steps, 1) / 100
    print(f"AE({F},{Z}) path {h.path}: encode {te:.3f} ms, decode {td:.3f} ms, fwd_bwd {tt:.3f} ms per 1M rows, bs512 step {1e3 * t5:.1f} us")
