#!/usr/bin/env python3
"""fp64 bamd_fwd_bwd of the fused pair by batch size, weight-gradient tiles in 2 x 4 blocks (default from 8,192 rows) vs one tile per
workgroup (BALER_AMD_DW64_MACRO_BLKS=0), and the difference of the two gradients:  gpurun -- python tools/bench_fp64_train.py [LIB.so]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
torch.manual_seed(1)
x = torch.rand((262144, 24), dtype=torch.float64, device="cuda")
out = []
import os
for n in [int(v) for v in os.environ.get('SIZES', '4096,8192,16384,65536,262144').split(',')]:
    g = torch.zeros_like(p)
    for _ in range(3): h.fwd_bwd(x[:n], g)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): h.fwd_bwd(x[:n], g)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    out.append("%%d: %%.3f ms = %%.3f of peak" %% (n, ms, 357000 * n / ms / 1e9 / 78.6))
    np.save("/tmp/f64g_%%s_%%d.npy" %% (sys.argv[1], n), g.cpu().numpy())
print("RES", " | ".join(out))
''' % R
lib = sys.argv[1] if len(sys.argv) > 1 else None
for name, tag, env in (("2x4 tile blocks", "m", {}), ("one tile per workgroup", "s", {"BALER_AMD_DW64_MACRO_BLKS": "0"})):
    e = dict(os.environ, **env)
    if lib:
        e["BALER_AMD_LIB"] = os.path.abspath(lib)
    o = subprocess.run([sys.executable, "-c", CHILD, tag], env=e, capture_output=True, text=True)
    print(f"{name:24s}", [l for l in o.stdout.splitlines() if l.startswith("RES")] or o.stderr[-600:])
import numpy as np
for n in [int(v) for v in os.environ.get('SIZES', '4096,8192,16384,65536,262144').split(',')]:
    a, b = np.load(f"/tmp/f64g_m_{n}.npy"), np.load(f"/tmp/f64g_s_{n}.npy")
    print(n, "gradient rel-L2 difference between the two kernels %.2e" % (np.linalg.norm(a - b) / np.linalg.norm(b)))
