import os, sys
sys.path.insert(0, '/root/repo')
import numpy as np, torch
from baler_amd import synth
from baler_amd.modules import models
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15); flat = orc.formula_params(dims, 7)
x = orc.normalize(synth.cms_rows(20000))
m = models.AE(24, 15, mode="bf16").load_flat(flat).to("cuda:0"); h = m.handle()
layout, _ = models.tensor_layout(dims)
n=272
g = torch.zeros_like(m.flat); h.fwd_bwd(torch.as_tensor(x[:n]).cuda(), g)
_, g_ref = orc.fwd_bwd(dims, flat, x[:n]); gh = g.cpu().numpy().astype(np.float64)
for key, off, shape in layout:
    k = int(np.prod(shape)); a=gh[off:off+k]; b=g_ref[off:off+k]
    print(f"{key:12s} rel {np.linalg.norm(a-b)/np.linalg.norm(b):.3e}  ratio {np.dot(a,b)/np.dot(b,b):.4f}")
