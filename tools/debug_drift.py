import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
g = np.load("tests/golden/g7_c1_cli.npz")
dims = orc.ae_dims(24, 15)
init = orc.formula_params(dims, int(g["init_seed"]))
data = torch.as_tensor(orc.normalize(synth.cms_rows(10000))).cuda()
h = native.Handle(dims, "fp32")
p = torch.as_tensor(np.concatenate([init, [0.0]])).float().cuda()
h.load_params(p)
m, v, grads = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
acc = torch.zeros(1, dtype=torch.float64, device="cuda")
t = 0
devs = []
for ep in range(12):
    acc.zero_()
    nb = 0
    for s in range(0, 10000, 512):
        h.fwd_bwd(data[s:s + 512], grads)
        t += 1
        h.adam_step(p, grads, m, v, t, 1e-3, loss_accum=acc)
        nb += 1
    el = acc.item() / nb
    devs.append(abs(el / g["loss_data"][0][ep] - 1))
print("latency_rows", os.environ.get("BALER_AMD_LATENCY_ROWS", "default"), " ".join(f"{d:.1e}" for d in devs))
