#!/usr/bin/env python3
"""Large training batches of a 64..127-column table: layer-wise (default) against the small-batch kernels taking the whole batch
(BALER_AMD_LATENCY_ROWS at handle creation), GPU box: python tools/bench_mid_width_train.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from baler_amd import native
from oracle import c_oracle as orc

os.environ["BALER_AMD_QUIET"] = "1"
for F, Z in ((80, 16), (64, 16), (127, 31)):
    dims = orc.ae_dims(F, Z)
    for rows in (65536, 262144, 1000000):
        x = torch.rand((rows, F), dtype=torch.float64, device="cuda")
        for lat in ("12288", "4000000"):
            os.environ["BALER_AMD_LATENCY_ROWS"] = lat
            h = native.Handle(dims, "fp32")
            os.environ.pop("BALER_AMD_LATENCY_ROWS")
            p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
            h.load_params(p)
            g = torch.zeros_like(p)
            for _ in range(3):
                h.fwd_bwd(x, g)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5):
                h.fwd_bwd(x, g)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5
            print(f"AE({F},{Z}) {rows} rows, small-batch limit {lat}: fwd_bwd {t:.3f} ms = {rows / t / 1e3:.1f} M rows/s", flush=True)
            h.close()
        del x
