#!/usr/bin/env python3
"""C4 in the bf16 mode: CFD_dense_AE(2500, 25) encode / decode with en1 / de4 on the bf16 MFMA -- HBM-bound kernels
(10 KB of float32 per frame): frames/s and the fraction of the 8 TB/s HBM roof.  python tools/bench_c4_bf16.py [frames]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
x = torch.as_tensor(synth.cfd_field(n).reshape(n, 2500).astype(np.float32)).cuda()
dims = orc.ae_dims(2500, 25)
flat = orc.formula_params(dims, 1)
res = {}
for mode in ("fp32", "bf16"):
    h = native.Handle(dims, mode)
    h.load_params(torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda())
    def timeit(fn, k=5):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k
    z = h.encode(x, out_dtype=torch.float32)
    te = timeit(lambda: h.encode(x, out_dtype=torch.float32))
    td = timeit(lambda: h.decode(z))
    res[mode] = (z, h.decode(z))
    print(f"C4 {mode} N={n}: encode {n / te / 1e6:.1f} M frames/s ({(10000 + 100) * n / te / 1e12:.2f} TB/s of rows + latents = "
          f"{100 * 10100 * n / te / 8e12:.0f}% of HBM); decode {n / td / 1e6:.1f} M frames/s ({100 * 10100 * n / td / 8e12:.0f}% of HBM)")
e = lambda a, b: float(torch.linalg.norm(a.double() - b.double()) / torch.linalg.norm(b.double()))
print(f"bf16 vs fp32: encode rel-L2 {e(res['bf16'][0], res['fp32'][0]):.2e}, decode {e(res['bf16'][1], res['fp32'][1]):.2e}")
