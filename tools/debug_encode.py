import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
flat = orc.formula_params(dims, 51)
h = native.Handle(dims, "fp32")
p = torch.as_tensor(np.concatenate([flat, [0.0]])).float().cuda()
h.load_params(p)
n = 1_000_000
x = torch.as_tensor(orc.normalize(synth.cms_rows(n))).float().cuda()
z = h.encode(x)
z_again = h.encode(x)
print("deterministic:", torch.equal(z, z_again), float((z - z_again).abs().max()))
for cut in (333_331, 333_328, 16, 17, 64):
    z2 = torch.cat([h.encode(x[:cut]), h.encode(x[cut:])])
    d = (z - z2).abs()
    bad = (d > 0).any(dim=1).nonzero().flatten()
    print("cut", cut, "equal:", torch.equal(z, z2), "max", float(d.max()), "nbad rows", bad.numel(),
          bad[:8].tolist(), bad[-4:].tolist())
