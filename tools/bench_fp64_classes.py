#!/usr/bin/env python3
"""fp64 mode on narrow tables other than the 24-column one: class instantiations of the fp64 kernels against the layer-wise path
(BALER_AMD_FORCE_GENERIC=1 in a child process).  python tools/bench_fp64_classes.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
def ms(fn, reps=3):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(4):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / reps)
    return best
n = 262144
for F, Z in ((24, 15), (30, 8), (47, 12), (63, 15)):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "fp64")
    p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
    h.load_params(p)
    x = torch.rand((n, F), dtype=torch.float64, device="cuda")
    g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    st = {"t": 0}
    def steps():
        for i in range(50):
            st["t"] += 1
            h.train_step(x[i * 512:(i + 1) * 512], p, m, v, st["t"], 1e-3)
    print("AE(%%d,%%d) path %%s: encode %%.3f ms, fwd_bwd %%.3f ms per %%d rows, bs512 step %%.1f us" %% (F, Z, h.path, ms(lambda: h.encode(x)), ms(lambda: h.fwd_bwd(x, g)), n, 1e3 * ms(steps, 1) / 50))
''' % R
for tag, env in (("fused", {}), ("layer-wise", {"BALER_AMD_FORCE_GENERIC": "1"})):
    o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BALER_AMD_QUIET="1", **env), capture_output=True, text=True)
    print("==", tag); print(o.stdout.strip() or o.stderr[-600:])
