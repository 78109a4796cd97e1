#!/usr/bin/env python3
"""(Experiments of commits b012638 (BALER_AMD_BF16_ENC256) and beeb36f (BALER_AMD_BF16_ENC_REG, what this version toggles): the switches exist only there.)  bf16 encode of the wide models: 128-row groups (wide_bf16_encode_dma_kernel) against 256-row groups (wide_bf16_encode_dma256_kernel,
BALER_AMD_BF16_ENC256 = minimum rows, 0 = off), outputs compared."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
from _gpu_warm import warm
def ms(fn, reps):
    warm(40.0)
    for _ in range(5): fn()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[2]
for name, F, Z, ns in (("C4 2500-25", 2500, 25, (32768, 65536, 131072, 262144, 100001)), ("C5 512-6", 512, 6, (65536, 262144, 1048576, 300007))):
    dims = orc.ae_dims(F, Z)
    h = native.Handle(dims, "bf16")
    torch.manual_seed(0)
    p = (torch.randn(sum((dims[i] + 1) * dims[i + 1] for i in range(8)) + 1) * 0.05).float().cuda()
    h.load_params(p)
    for n in ns:
        x = torch.rand((n, F), dtype=torch.float32, device="cuda")
        res = {}
        for tag, env in (("loader waves (DMA)", "0"), ("loaders, rows in chunk pairs", "1")):
            os.environ["BALER_AMD_BF16_ENC_PAIR"] = env
            z = torch.empty((n, Z), dtype=torch.float32, device="cuda")
            h.encode(x, out=z); torch.cuda.synchronize()
            reps = max(3, min(50, int(2e9 / (n * F * 4))))
            t = ms(lambda: h.encode(x, out=z), reps)
            res[tag] = (t, z.clone())
        gb = n * (F + Z) * 4 / 1e9
        d = max(float((res[k][1] - res["loader waves (DMA)"][1]).abs().max()) for k in res)
        print(f"{name} {n:8d} rows: " + "   ".join(f"{k} {v[0] * 1e3:8.1f} us = {gb / v[0] / 8:.3f} of HBM" for k, v in res.items()) + f"   max |diff| {d:.2e}", flush=True)
        del x
    h.close()
