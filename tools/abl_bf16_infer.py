#!/usr/bin/env python3
"""Interleaved A/B of bf16 encode / decode variants of AE(24, 15): python tools/abl_bf16_infer.py lib1.so lib2.so ...
(each library in its own process; median of 5 x 10 event-timed launches after 30 warm ones, 1M and 4M rows, float64 and float32 rows;
the last two numbers are checksums of z and of the reconstruction: variants must agree)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda())
out = []
torch.manual_seed(0)
for n in (1000000, 4000000):
    for dt in (torch.float64, torch.float32):
        x = torch.rand((n, 24), dtype=dt, device="cuda")
        z = h.encode(x); y = h.decode(z)
        for fn, tag in ((lambda: h.encode(x, out=z), "enc"), (lambda: h.decode(z, out=y), "dec")):
            for _ in range(30): fn()
            torch.cuda.synchronize()
            ts = []
            for _ in range(5):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(10): fn()
                e1.record(); torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) / 10)
            out.append("%%s%%dM%%s %%.1f us %%.2f G/s" %% (tag, n // 1000000, "f64" if dt == torch.float64 else "f32", 1e3 * sorted(ts)[2], n / sorted(ts)[2] / 1e6))
print("RES", " | ".join(out), "%%.6e %%.6e" %% (float(z.double().abs().sum()), float(y.double().abs().sum())))
''' % R
for rnd in range(2):
    for l in sys.argv[1:]:
        o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BALER_AMD_LIB=os.path.abspath(l)), capture_output=True, text=True, timeout=600)
        line = [x for x in o.stdout.splitlines() if x.startswith("RES")]
        print(f"{os.path.basename(l):16s}", line[0][4:] if line else o.stderr[-600:], flush=True)
