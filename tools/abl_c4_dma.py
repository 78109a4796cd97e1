#!/usr/bin/env python3
"""Timing-only ablation builds of wide_bf16_encode_dma_kernel (-DBAMD_DMA_ABL=1..4, .abl/libdma*.so), each in its own process:
which part of a chunk iteration holds the kernel below the loaders' own rate.  python tools/abl_c4_dma.py [frames]"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
n = int(sys.argv[1]) if len(sys.argv) > 1 else 131072
CHILD = r'''
import sys; sys.path.insert(0, %r)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
n = %d
import os
x = torch.zeros((n, 2500), dtype=torch.float32, device="cuda") if os.environ.get("ZERO") else torch.rand((n, 2500), dtype=torch.float32, device="cuda")
dims = orc.ae_dims(2500, 25)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda())
for _ in range(40): z = h.encode(x, out_dtype=torch.float32)
ts = []
for _ in range(5):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10): h.encode(x, out_dtype=torch.float32)
    e1.record(); torch.cuda.synchronize()
    ts.append(e0.elapsed_time(e1) / 10)
t = sorted(ts)[2]
print("RES %%.4f ms = %%.1f M frames/s = %%.2f TB/s" %% (t, n / t / 1e3, 10100.0 * n / t / 1e9))
''' % (R, n)
import glob
libs = [("full kernel", os.path.join(R, "baler_amd", "libbaler_amd.so"))] + [(os.path.basename(l), l) for l in sorted(glob.glob(os.path.join(R, ".abl", "lib*.so")))]
for rnd in range(2):
    for name, l in libs:
        o = subprocess.run([sys.executable, "-c", CHILD], env=dict(os.environ, BALER_AMD_LIB=l), capture_output=True, text=True, timeout=600)
        line = [x for x in o.stdout.splitlines() if x.startswith("RES")]
        print(f"{name:52s}", line[0][4:] if line else o.stderr[-400:], flush=True)
