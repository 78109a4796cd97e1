import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
flat = orc.formula_params(dims, 14)
x = orc.normalize(synth.cms_rows(10000))[:512]
lo, go = orc.fwd_bwd(dims, flat, x)
h = native.Handle(dims, "fp32")
p = torch.as_tensor(np.concatenate([flat, [0.0]])).float().cuda()
h.load_params(p)
grads = torch.zeros_like(p)
h.fwd_bwd(torch.as_tensor(x).cuda(), grads)
gh = grads.cpu().numpy().astype(np.float64)
# oracle evaluated at the fp32-rounded parameters (separates kernel error from parameter rounding)
lo32, go32 = orc.fwd_bwd(dims, flat.astype(np.float32).astype(np.float64), x)
names = ["en1", "en2", "en3", "en4", "de1", "de2", "de3", "de4"]
off = 0
out = []
for l in range(8):
    for kind, n in (("W", dims[l + 1] * dims[l]), ("b", dims[l + 1])):
        a, b = gh[off:off + n], go32[off:off + n]
        out.append(f"{names[l]}.{kind}:{np.linalg.norm(a - b) / np.linalg.norm(b):.1e}")
        off += n
print(os.environ.get("BALER_AMD_LATENCY_ROWS", "lat"), os.environ.get("BALER_AMD_FORCE_GENERIC", ""), " ".join(out), f"loss:{abs(gh[-1]-lo32)/lo32:.1e}")
