#!/usr/bin/env python3
"""4-row chain (lat4_chain_kernel) vs the 16-row chain and the oracle, per tensor:  gpurun -- python tools/check_lat4.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from baler_amd.modules import models
from oracle import c_oracle as orc

z = int(sys.argv[1]) if len(sys.argv) > 1 else 15
dims = orc.ae_dims(24, z)
flat = orc.formula_params(dims, 7)
x = orc.normalize(synth.cms_rows(1000))
layout, _ = models.tensor_layout(dims)
def grads(env, n):
    os.environ["BALER_AMD_LAT4_ROWS"] = env
    h = native.Handle(dims, "fp32")
    p = torch.as_tensor(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    g = torch.zeros_like(p)
    h.fwd_bwd(torch.as_tensor(x[:n]).cuda(), g)
    return g.cpu().numpy().astype(np.float64)
for n in (4, 16, 272, 1000):
    g4, g2 = grads("1024", n), grads("0", n)
    lo, go = orc.fwd_bwd(dims, flat, x[:n])
    print(f"n={n}: loss lat4 {g4[-1]:.6f} lat2 {g2[-1]:.6f} oracle {lo:.6f}")
    for key, off, shape in layout:
        k = int(np.prod(shape))
        e4 = np.linalg.norm(g4[off:off + k] - go[off:off + k]) / max(np.linalg.norm(go[off:off + k]), 1e-300)
        e2 = np.linalg.norm(g2[off:off + k] - go[off:off + k]) / max(np.linalg.norm(go[off:off + k]), 1e-300)
        print(f"   {key:12s} lat4 {e4:.2e}  lat2 {e2:.2e}")
