#!/usr/bin/env python3
"""fp64 encode / decode / forward + loss of 64 .. 127-column tables: the register-chained inference kernel (fused64j.hip) against the
layer-wise kernels (BALER_AMD_F64_INFER=0, read once per process).  python tools/bench_fp64_mid_width_infer.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
n = 262144
for F, Z in ((80, 16), (127, 31), (100, 63)):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "fp64")
    h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda())
    x = torch.rand((n, F), dtype=torch.float64, device="cuda")
    z = h.encode(x); y = torch.empty_like(x)
    out = []
    for tag, fn in (("encode", lambda: h.encode(x, out=z)), ("decode", lambda: h.decode(z, out=y)), ("forward+loss", lambda: h.forward_loss(x, want_recon=False))):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        out.append("%%s %%.3f ms = %%.0f M rows/s" %% (tag, e0.elapsed_time(e1) / 10, n / (e0.elapsed_time(e1) / 10) / 1e3))
    print("RES AE(%%d,%%d) %%s: %%s" %% (F, Z, h.path, " | ".join(out)))
    h.close()
''' % R
for name, env in (("register chain (fused64j.hip)", {}), ("layer-wise", {"BALER_AMD_F64_INFER": "0"})):
    o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BALER_AMD_QUIET="1", **env), capture_output=True, text=True)
    print(name)
    for l in o.stdout.splitlines():
        if l.startswith("RES"): print("  ", l[4:])
    if o.returncode: print(o.stderr[-500:])
