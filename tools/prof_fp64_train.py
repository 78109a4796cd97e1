import sys; sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
n = int(sys.argv[1])
x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
g = torch.zeros_like(p)
for _ in range(8): h.fwd_bwd(x, g)
torch.cuda.synchronize(); print("done")
