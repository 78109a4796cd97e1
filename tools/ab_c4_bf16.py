#!/usr/bin/env python3
"""A/B of the bf16 encode of the wide models in ONE process: the kernel with the decoupled row stream (loader waves +
LDS ring, default) against the register-streamed kernel (BALER_AMD_WIDE_DMA=0), interleaved rounds, median and best
(cdna_hip_programming.md rule 24), results compared.  python tools/ab_c4_bf16.py [frames] [cols] [latent]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
F = int(sys.argv[2]) if len(sys.argv) > 2 else 2500
Z = int(sys.argv[3]) if len(sys.argv) > 3 else 25
x = torch.rand((n, F), dtype=torch.float32, device="cuda")
dims = orc.ae_dims(F, Z)
flat = orc.formula_params(dims, 1)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda())
out = {}
def run(v, reps=10):
    os.environ["BALER_AMD_WIDE_DMA"] = v
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        z = h.encode(x, out_dtype=torch.float32)
    e1.record()
    torch.cuda.synchronize()
    out[v] = z
    return e0.elapsed_time(e1) / reps
for v in ("1", "0"):
    run(v, 30)
t = {"1": [], "0": []}
for _ in range(7):
    for v in ("1", "0"):
        t[v].append(run(v))
bytes_row = F * 4 + Z * 4
for v, name in (("1", "loader waves + LDS ring"), ("0", "rows through registers")):
    med, best = float(np.median(t[v])), min(t[v])
    print(f"{name:28s}: median {med:.4f} ms = {n / med / 1e3:.1f} M rows/s = {bytes_row * n / med / 1e9:.2f} TB/s = "
          f"{100 * bytes_row * n / med / 8e12:.1f} % of 8 TB/s   (best {best:.4f} ms)")
a, b = out["1"].double(), out["0"].double()
print("results: rel-L2 between the two kernels", float(torch.linalg.norm(a - b) / torch.linalg.norm(b)), "finite", bool(torch.isfinite(a).all()))
m = 512
zr = orc.encode(dims, flat, x[:m].cpu().numpy().astype(np.float64))
print("DMA kernel vs fp64 oracle on the first", m, "rows: rel-L2", float(np.linalg.norm(out["1"][:m].cpu().numpy() - zr) / np.linalg.norm(zr)))
zt = orc.encode(dims, flat, x[-m:].cpu().numpy().astype(np.float64))
print("... and the last", m, "rows:", float(np.linalg.norm(out["1"][-m:].cpu().numpy() - zt) / np.linalg.norm(zt)))
