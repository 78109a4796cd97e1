"""Small-batch kernels vs the C oracle: gradient, loss and a few fused Adam steps (debug helper)."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baler_amd import native, synth
from oracle import c_oracle as orc

dims = orc.ae_dims(24, 15)
p0 = orc.formula_params(dims, 3)
for n in (512, 16, 272, 1000, 4096, 37):
    x = synth.cms_rows(n)
    xn = orc.normalize(x)
    loss_ref, g_ref = orc.fwd_bwd(dims, p0, xn)
    h = native.Handle(dims, "fp32")
    flat = torch.from_numpy(p0.astype(np.float32)).cuda()
    h.load_params(flat)
    xd = torch.from_numpy(xn).cuda()
    grads = torch.zeros(h.nparams + 1, dtype=torch.float32, device="cuda")
    h.fwd_bwd(xd, grads)
    g = grads.cpu().numpy().astype(np.float64)
    err = np.abs(g[:-1] - g_ref).max() / np.abs(g_ref).max()
    print(n, "grad rel err", err, "loss rel", abs(g[-1] - loss_ref) / loss_ref)
    # fused steps vs fwd_bwd + adam
    m = torch.zeros_like(flat); v = torch.zeros_like(flat); la = torch.zeros(1, dtype=torch.float64, device="cuda")
    h2 = native.Handle(dims, "fp32"); flat2 = flat.clone(); h2.load_params(flat2)
    m2 = torch.zeros_like(flat); v2 = torch.zeros_like(flat); la2 = torch.zeros(1, dtype=torch.float64, device="cuda")
    for t in range(1, 6):
        h.fwd_bwd(xd, grads); h.adam_step(flat, grads, m, v, t, 1e-3, loss_accum=la)
        h2.train_step(xd, flat2, m2, v2, t, 1e-3, loss_accum=la2)
    torch.cuda.synchronize()
    print("   fused vs split: params equal", bool(torch.equal(flat, flat2)), "m", bool(torch.equal(m, m2)), "loss", la.item(), la2.item())
    z1 = h.encode(xd); z2 = h2.encode(xd)
    print("   packed equal (encode)", bool(torch.equal(z1, z2)))
