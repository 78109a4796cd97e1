#!/usr/bin/env python3
"""fp64 encode / decode / forward rate of the register-chained kernel vs the layer-wise path: gpurun -- python tools/bench_fp64_infer.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, time; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
n = 1 << 20
x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
z = h.encode(x)
def ms(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps
te, td, tf = ms(lambda: h.encode(x)), ms(lambda: h.decode(z)), ms(lambda: h.forward_loss(x, want_recon=False))
print("RES encode %%.3f ms = %%.1f TF (%%.0f%%%% of 78.6)  decode %%.3f ms = %%.1f TF  forward+loss %%.3f ms = %%.1f TF" %% (
    te, 61100 * n / te / 1e9, 61100 * n / te / 1e9 / 0.786, td, 61100 * n / td / 1e9, tf, 122200 * n / tf / 1e9))
''' % R
for name, env in (("register chain, 512 WGs", {"BALER_AMD_F64_INFER_WGS": "512"}), ("register chain, 256 WGs", {"BALER_AMD_F64_INFER_WGS": "256"}),
                  ("layer-wise", {"BALER_AMD_F64_INFER": "0"})):
    o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, **env), capture_output=True, text=True)
    print(f"{name:26s}", [l for l in o.stdout.splitlines() if l.startswith("RES")] or o.stderr[-400:])
