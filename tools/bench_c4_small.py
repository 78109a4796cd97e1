import os, sys, time
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(2500, 25)
flat = orc.formula_params(dims, 1)
h = native.Handle(dims, "fp32")
p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
g = torch.zeros_like(p)
for n in (60, 512, 2048, 6000):
    x = torch.rand((n, 2500), dtype=torch.float32, device="cuda")
    def t(fn, k=20):
        fn(); torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(k): fn()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / k * 1e6
    res = []
    for wt in ("1", "0"):
        os.environ["BALER_AMD_WIDE_TRAIN"] = wt
        res.append(t(lambda: h.fwd_bwd(x, g)))
    te = t(lambda: h.encode(x)); td = t(lambda: h.decode(h.encode(x))) - te
    os.environ["BALER_AMD_FORCE_GENERIC"] = "1"
    hg = native.Handle(dims, "fp32"); hg.load_params(p)
    del os.environ["BALER_AMD_FORCE_GENERIC"]
    tge = t(lambda: hg.encode(x))
    print(f"n={n}: fwd_bwd fused {res[0]:.0f} us, layer-wise {res[1]:.0f} us; encode fused {te:.0f} us, layer-wise {tge:.0f} us; decode fused {td:.0f} us")
