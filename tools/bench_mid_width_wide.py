#!/usr/bin/env python3
"""64..127-column tables: the two-state handle (wide class for inference and large batches + small-batch class; default) against the
small-batch class alone (BALER_AMD_MID_HYBRID=0: large batches chunked on its kernels).  GPU box: python tools/bench_mid_width_wide.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
os.environ["BALER_AMD_QUIET"] = "1"
for F, Z in ((80, 16), (64, 16), (100, 1), (127, 31)):
    dims = orc.ae_dims(F, Z)
    for rows in (65536, 1000000):
        x = torch.rand((rows, F), dtype=torch.float64, device="cuda")
        for exp in (False, True):
            if exp: os.environ.pop("BALER_AMD_MID_HYBRID", None)
            else: os.environ["BALER_AMD_MID_HYBRID"] = "0"
            h = native.Handle(dims, "fp32")
            p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
            h.load_params(p)
            g = torch.zeros_like(p)
            for _ in range(3): h.fwd_bwd(x, g)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(5): h.fwd_bwd(x, g)
            e1.record(); torch.cuda.synchronize()
            t = e0.elapsed_time(e1) / 5
            z = h.encode(x); torch.cuda.synchronize()
            e0.record()
            for _ in range(5): z = h.encode(x)
            e1.record(); torch.cuda.synchronize()
            te = e0.elapsed_time(e1) / 5
            m, v = torch.zeros_like(p), torch.zeros_like(p)
            xs = x[:512]
            for k in range(20): h.train_step(xs, p, m, v, k + 1, 1e-3)
            torch.cuda.synchronize()
            e0.record()
            for k in range(200): h.train_step(xs, p, m, v, k + 21, 1e-3)
            e1.record(); torch.cuda.synchronize()
            ts = e0.elapsed_time(e1) / 200 * 1e3
            print(f"AE({F},{Z}) {rows} rows, two_state={exp}: fwd_bwd {t:.3f} ms = {rows / t / 1e3:.1f} M rows/s; encode {te:.3f} ms; 512-row step {ts:.1f} us", flush=True)
            h.close()
        del x
