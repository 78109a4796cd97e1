#!/usr/bin/env python3
"""encode with / without normalise-on-load (features), by mode and row dtype:  gpurun -- python tools/bench_norm_on_load.py [ROWS]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000000
dims = orc.ae_dims(24, 15)
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
torch.manual_seed(0)
raw64 = torch.rand((n, 24), dtype=torch.float64, device="cuda") * 37.0 - 5.0
for mode in ("bf16", "fp32"):
    h = native.Handle(dims, mode)
    h.load_params(p.float() if mode != "fp64" else p)
    for dt in (torch.float64, torch.float32):
        raw = raw64.to(dt)
        feats = native.minmax(raw)
        xn = native.normalize(raw, feats)
        res = []
        for name, fn in (("pre-normalised", lambda: h.encode(xn)), ("normalise-on-load", lambda: h.encode(raw, features=feats))):
            for _ in range(5): fn()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20): fn()
            e1.record(); torch.cuda.synchronize()
            res.append(f"{name} {n / (e0.elapsed_time(e1) / 20) / 1e6:.2f} G rows/s")
        za, zb = h.encode(xn), h.encode(raw, features=feats)
        res.append("max |dz| %.3e" % float((za.double() - zb.double()).abs().max()))
        print(mode, str(dt).split(".")[-1], " | ".join(res))
