import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
os.environ["BALER_AMD_QUIET"]="1"
shape=(2500,25); n=1000
dims = orc.ae_dims(*shape)
flat = orc.formula_params(dims, 41)
x = np.random.default_rng(n).random((n, shape[0]))
lo, go = orc.fwd_bwd(dims, flat, x)
def run():
    h = native.Handle(dims, "fp32")
    p = torch.from_numpy(np.concatenate([flat,[0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    g = torch.zeros_like(p)
    h.fwd_bwd(torch.as_tensor(x, dtype=torch.float32).cuda(), g)
    return g.cpu().numpy().astype(np.float64)
ga = run()
print("small path: relL2", np.linalg.norm(ga[:-1]-go)/np.linalg.norm(go), "max-norm", np.abs(ga[:-1]-go).max()/np.abs(go).max(), "loss", abs(ga[-1]-lo)/lo)
off=0
for l in range(8):
    for cnt in (dims[l+1]*dims[l], dims[l+1]):
        a,b=ga[off:off+cnt],go[off:off+cnt]
        print(l, cnt, "relL2 %.2e max %.2e"%(np.linalg.norm(a-b)/np.linalg.norm(b), np.abs(a-b).max()/np.abs(b).max()))
        off+=cnt
