#!/usr/bin/env python3
"""fp64 optimiser step (bamd_train_step, BAMD_MODE_F64) against the batch size, 4-row chain (chain64q_kernel) vs 16-row exchange chain
(chain64_kernel): us per step over 200 back-to-back steps.  python tools/bench_fp64_small_steps.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from baler_amd import native, synth
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
m, v = torch.zeros_like(p), torch.zeros_like(p)
x = torch.from_numpy(orc.normalize(synth.cms_rows(8192))).cuda()
print("rows   four-row chain   exchange chain   (us per bamd_train_step)")
for n in ([int(a) for a in sys.argv[1:]] or [16, 64, 256, 512, 768, 1024, 1536, 2048, 4096]):
    row = []
    for q in ("1000000", "0"):
        os.environ["BALER_AMD_F64_QCHAIN_BLKS"] = q
        t = 0
        for _ in range(50):
            t += 1; h.train_step(x[:n], p, m, v, t, 1e-3)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(200):
            t += 1; h.train_step(x[:n], p, m, v, t, 1e-3)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 200 * 1e6)
    print(f"{n:5d}   {row[0]:10.1f}      {row[1]:10.1f}", flush=True)
