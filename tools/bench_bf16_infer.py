import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch, time
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda())
for dt in (torch.float64, torch.float32):
    x = torch.rand((4_000_000, 24), dtype=dt, device="cuda")
    z = h.encode(x)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): z = h.encode(x)
    torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 10
    print(dt, f"encode {4e6 / dt_ / 1e9:.2f} G rows/s")
    y = h.decode(z)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): y = h.decode(z)
    torch.cuda.synchronize(); dt_ = (time.perf_counter() - t0) / 10
    print(dt, f"decode {4e6 / dt_ / 1e9:.2f} G rows/s")
