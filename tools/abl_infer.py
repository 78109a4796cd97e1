#!/usr/bin/env python3
"""Interleaved A/B of fp32 encode / decode variants: python tools/abl_infer.py lib1.so lib2.so ...  (each in its own process;
median of event-timed launches at 1M and 4M float64 rows)."""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import torch
from baler_amd import native, synth
from baler_amd.modules import models
torch.manual_seed(0)
m = models.AE(24, 15, mode="fp32").to("cuda:0")
h = m.handle()
out = []
for n in (1000000, 4000000):
    x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
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
        out.append("%%s%%dM %%.4f ms = %%.3f G rows/s" %% (tag, n // 1000000, sorted(ts)[2], n / sorted(ts)[2] / 1e6))
print("RES", " | ".join(out), float(z.double().abs().sum()), float(y.double().abs().sum()))
''' % R
for rnd in range(2):
    for l in sys.argv[1:]:
        o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BALER_AMD_LIB=os.path.abspath(l)), capture_output=True, text=True, timeout=600)
        line = [x for x in o.stdout.splitlines() if x.startswith("RES")]
        print(f"{os.path.basename(l):16s}", line[0][4:] if line else o.stderr[-400:])
