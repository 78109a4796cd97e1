#!/usr/bin/env python3
"""fp64 optimiser step of 64 .. 127-column tables at the reference's batch sizes: the 4-row chain (Impl64Q: chain64q_kernel + dw64_kernel)
against the layer-wise kernels (BALER_AMD_F64_QCHAIN_BLKS=0, read per call).  python tools/bench_fp64_mid_width_step.py"""
import os, sys; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch, time
from baler_amd import native
from oracle import c_oracle as orc
os.environ["BALER_AMD_QUIET"] = "1"
print("model        rows   fused (4-row chain)   layer-wise   (us per bamd_train_step, fp64)")
for F, Z in ((64, 16), (80, 16), (100, 31), (127, 31)):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "fp64")
    p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
    h.load_params(p)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    x = torch.rand((65536, F), dtype=torch.float64, device="cuda")
    for n in ([int(a) for a in sys.argv[1:]] or [64, 512, 1536]):
        row = []
        for q in (None, "0"):
            if q is None: os.environ["BALER_AMD_F64_QCHAIN_BLKS"] = "1000000"
            else: os.environ["BALER_AMD_F64_QCHAIN_BLKS"] = q
            t = 0
            for _ in range(10):
                t += 1; h.train_step(x[:n], p, m, v, t, 1e-3)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(20):
                t += 1; h.train_step(x[:n], p, m, v, t, 1e-3)
            torch.cuda.synchronize()
            row.append((time.perf_counter() - t0) / 20 * 1e6)
        os.environ.pop("BALER_AMD_F64_QCHAIN_BLKS", None)
        print(f"AE({F:3d},{Z:2d})  {n:5d}   {row[0]:12.1f}        {row[1]:10.1f}", flush=True)
    h.close()
