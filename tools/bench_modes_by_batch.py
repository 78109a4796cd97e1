import sys, time; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
flat = orc.formula_params(dims, 1)
x = torch.rand((262144, 24), dtype=torch.float64, device="cuda")
for mode in ("fp32", "bf16"):
    h = native.Handle(dims, mode)
    p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    for bs in (512, 2048, 8192, 32768):
        nb = min(200, x.shape[0] // bs)
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for i in range(nb): h.train_step(x[i * bs:(i + 1) * bs], p, m, v, i + 1, 1e-3)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / nb
        print(f"{mode} train_step {bs} rows: {dt * 1e6:.1f} us/step = {bs / dt / 1e6:.1f} M rows/s")
