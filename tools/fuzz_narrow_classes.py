#!/usr/bin/env python3
"""Randomised parity sweep of the class instantiations: random (columns, latent) pairs up to 79 x 31 and random row counts;
encode / decode / forward + loss / fwd_bwd (small-batch and throughput sizes) of the fp32 kernels against the fp64 oracle.
python tools/fuzz_narrow_classes.py [shapes] [seed]      (GPU box; the oracle is the checker here, as in tests/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc

def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    return max(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300), np.abs(a - b).max() / max(np.abs(b).max(), 1e-300))

count, seed = (int(sys.argv[1]) if len(sys.argv) > 1 else 40), (int(sys.argv[2]) if len(sys.argv) > 2 else 1)
rng = np.random.default_rng(seed)
worst, fails = 0.0, 0
os.environ["BALER_AMD_QUIET"] = "1"
for k in range(count):
    F, Z = int(rng.integers(1, 80)), int(rng.integers(1, 32))
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 1000 + k)
    h = native.Handle(dims, "fp32")
    p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    errs = {}
    for n in (int(rng.integers(1, 600)), int(rng.integers(12000, 14000))):
        x = rng.random((n, F))
        xd = torch.as_tensor(x).cuda()
        z_ref = orc.encode(dims, flat, x)
        errs[f"enc{n}"] = rel(h.encode(xd).cpu().numpy(), z_ref)
        errs[f"dec{n}"] = rel(h.decode(torch.as_tensor(z_ref).cuda()).cpu().numpy(), orc.decode(dims, flat, z_ref))
        recon, loss = h.forward_loss(xd)
        want = orc.forward(dims, flat, x)
        errs[f"fwd{n}"] = rel(recon.cpu().numpy(), want)
        lo, go = orc.fwd_bwd(dims, flat, x)
        g = torch.zeros_like(p)
        h.fwd_bwd(xd, g)
        gh = g.cpu().numpy().astype(np.float64)
        errs[f"grad{n}"] = np.linalg.norm(gh[:-1] - go) / np.linalg.norm(go)       # rel-L2: random rows may sit on a LeakyReLU kink
        errs[f"loss{n}"] = abs(gh[-1] - lo) / max(lo, 1e-300)
    w = max(errs.values())
    worst = max(worst, w)
    bad = {k_: v for k_, v in errs.items() if not v < 2e-5}
    fails += bool(bad)
    print(f"AE({F},{Z}) path {h.path:11s} worst {w:.2e}" + (f"  FAIL {bad}" if bad else ""), flush=True)
    h.close()
print(f"{count} shapes, worst {worst:.2e}, {fails} failing")
sys.exit(1 if fails else 0)
