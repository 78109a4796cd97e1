#!/usr/bin/env python3
"""C4 (CFD_dense_AE(2500, 25), 32,768 frames) for rocprofv3 with WARMED launches: a 150-ms clock warm-up (torch matmuls) + 100 launches of every entry point the
bench line quotes a fraction for (fp32 encode / decode / fwd_bwd; bf16 encode / decode / fwd_bwd), so that the per-kernel averages in
profiles/ reproduce `other_configs.c4_cfd_dense_2500_25` (round-5 review, weak #8: 4 cold launches did not)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import synth
from baler_amd.modules import models
from _gpu_warm import warm
n = int(sys.argv[1]) if len(sys.argv) > 1 else 32768
x = torch.as_tensor(synth.cfd_field(n).reshape(n, 2500).astype(np.float32)).cuda()
for mode in ("fp32", "bf16"):
    torch.manual_seed(0)
    m = models.CFD_dense_AE(2500, 25, mode=mode).to("cuda:0")
    h = m.handle()
    g = torch.zeros_like(m.flat)
    z = h.encode(x, out_dtype=torch.float32)
    y = torch.empty_like(x)
    for fn in (lambda: h.encode(x, out=z), lambda: h.decode(z, out=y), lambda: h.fwd_bwd(x, g)):
        warm(150.0)
        for _ in range(100):
            fn()
        torch.cuda.synchronize()
    h.close()
print("done")
