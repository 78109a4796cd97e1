"""Per-phase shader-clock timeline of the bf16 training kernels' workgroup 0 / wave 0, LAST iteration (needs a -DBAMD_BF16_TRACE
build of bf16_train.hip linked into an alternative library):
    cd baler_amd/csrc && mkdir -p ../../.abl && hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off \
        -mllvm -amdgpu-mfma-vgpr-form=1 -DBAMD_BF16_TRACE -c bf16_train.hip -o /tmp/bt_tr.o && \
      hipcc --offload-arch=gfx950 -shared -fPIC -o ../../.abl/btrace.so api.o elementwise.o generic.o fused.o fused64.o swd.o bf16.o /tmp/bt_tr.o
    BALER_AMD_LIB=$PWD/.abl/btrace.so python tools/bf16_trace.py          (on the GPU box)"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baler_amd import native, synth                               # noqa: E402
from baler_amd.modules import models                              # noqa: E402

x = torch.from_numpy(synth.cms_rows(256 * 64 * 8)).cuda()
x = native.normalize(x, native.minmax(x))
model = models.AE(24, 15, mode="bf16").to("cuda:0")
h = model.handle()
g = torch.zeros_like(model.flat)
for _ in range(3):
    h.fwd_bwd(x, g)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 96)()
L = native.lib()
L.bamd_debug_bf16_trace.argtypes = [ctypes.c_void_p]
print("rc", L.bamd_debug_bf16_trace(buf))
t = np.array(buf[:], dtype=np.int64).reshape(2, 48)
for part in (0, 1):
    tt = t[part]
    print(f"PART {part}: iteration total {tt[21 + 2 * (4 if part == 0 else 0)] - tt[0]} cycles")
    print(f"  rows -> image 0 + barrier   {tt[1] - tt[0]:6d}")
    prev = tt[1]
    for l in range(8 if part == 0 else 3):
        end = tt[9] if l == 7 else tt[2 + l]
        print(f"  forward layer {l}{' + loss' if l == 7 else '       '}     {end - prev:6d}")
        prev = end
    if part == 0:
        prev = tt[9]                              # forward layer 7 above includes the loss epilogue
    else:
        prev = tt[2 + 2]
    for l in (range(7, 3, -1) if part == 0 else range(3, -1, -1)):
        a, b = tt[20 + 2 * l], tt[21 + 2 * l]
        print(f"  backward layer {l}: chain + epilogue {a - prev:6d}   weight-gradient tiles + barrier {b - a:6d}")
        prev = b
