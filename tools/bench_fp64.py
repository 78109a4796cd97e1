#!/usr/bin/env python3
"""fp64 mode at the reference's batch size: us per bamd_train_step(512 rows), fused fp64 step vs the layer-wise kernels
(BALER_AMD_FORCE_GENERIC=1).  python tools/bench_fp64.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from baler_amd import native, synth
from baler_amd.modules import models

raw = torch.as_tensor(synth.cms_rows(512 * 200)).cuda()
x = native.normalize(raw, native.minmax(raw))
torch.manual_seed(0)
m = models.AE(24, 15, mode="fp64").to("cuda:0")
h = m.handle()
mm, vv = torch.zeros_like(m.flat), torch.zeros_like(m.flat)
for bs in (512, 4096):
    nb = min(200, x.shape[0] // bs)
    for rep in range(2):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(nb):
            h.train_step(x[i * bs:(i + 1) * bs], m.flat, mm, vv, i + 1, 1e-3)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / nb
    print(f"fp64 train_step, {bs} rows: {dt * 1e6:.1f} us/step = {bs / dt / 1e6:.2f} M rows/s "
          f"({'layer-wise' if os.environ.get('BALER_AMD_FORCE_GENERIC') == '1' else 'fused fp64'})")
