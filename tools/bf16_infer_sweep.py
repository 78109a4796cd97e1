#!/usr/bin/env python3
"""bf16 encode / decode of AE(24, 15): time against the row count for every (row dtype, output dtype) pair -- separates the per-row
cost from the per-launch cost (python tools/bf16_infer_sweep.py; median of 20 launches timed one by one with events, 20 warm ones)."""
import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda())
def t(fn):
    for _ in range(20): fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    return sorted(ts)[10]
ns = [int(a) for a in sys.argv[1:]] or [65536, 262144, 1000000, 2000000, 4000000, 8000000]
print("rows      " + "  ".join(f"{k:>14s}" for k in ("enc f64>f64", "enc f64>f32", "enc f32>f64", "enc f32>f32", "dec f64>f64", "dec f64>f32", "dec f32>f64", "dec f32>f32")))
for n in ns:
    row = []
    for kind in ("enc", "dec"):
        for din in (torch.float64, torch.float32):
            for dout in (torch.float64, torch.float32):
                x = torch.rand((n, 24 if kind == "enc" else 15), dtype=din, device="cuda")
                o = torch.empty((n, 15 if kind == "enc" else 24), dtype=dout, device="cuda")
                fn = (lambda: h.encode(x, out=o)) if kind == "enc" else (lambda: h.decode(x, out=o))
                row.append(t(fn))
                del x, o
    print(f"{n:9d} " + "  ".join(f"{v:8.1f} us   " for v in row), flush=True)
