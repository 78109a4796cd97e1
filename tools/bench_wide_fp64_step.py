import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
os.environ["BALER_AMD_QUIET"]="1"
for F,Z,rows in ((2500,25,60),(625,7,32),(2500,25,6000)):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "fp64")
    p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
    h.load_params(p); m, v = torch.zeros_like(p), torch.zeros_like(p)
    x = torch.rand((rows * 4, F), dtype=torch.float64, device="cuda")
    for rep in range(2):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for i in range(20): h.train_step(x[(i % 4) * rows:(i % 4 + 1) * rows], p, m, v, i + 1, 1e-3)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
    print(f"fp64 AE({F},{Z}) [{h.path}] train_step {rows} rows: {dt * 1e6:.1f} us/step")
