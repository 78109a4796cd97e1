import os, sys
sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
dims = orc.ae_dims(625, 7)
h = native.Handle(dims, "bf16")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
h.load_params(p)
x = torch.rand((131072, 625), dtype=torch.float32, device="cuda")
g = torch.zeros_like(p)
for _ in range(6):
    h.fwd_bwd(x, g)
torch.cuda.synchronize()
print("done")
