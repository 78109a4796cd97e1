"""Per-layer shader-clock timeline of the small-batch kernels' workgroup 0 (needs a -DBAMD_LAT_TRACE build):

    cd baler_amd/csrc && mkdir -p ../../.abl && \
      hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 \
            -DBAMD_LAT_TRACE -c fused.hip -o /tmp/fused_trace.o && \
      hipcc --offload-arch=gfx950 -shared -fPIC -o ../../.abl/trace.so api.o elementwise.o generic.o swd.o bf16.o /tmp/fused_trace.o
    BALER_AMD_LIB=$PWD/.abl/trace.so python tools/lat_trace.py [ROWS]   (on the GPU box)
"""
import ctypes
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baler_amd import native, synth                               # noqa: E402
from baler_amd.modules import models                              # noqa: E402

dev = torch.device("cuda", 0)
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512          # rows per step
x = torch.from_numpy(synth.cms_rows(R * 50)).to(dev)
x = native.normalize(x, native.minmax(x))
model = models.AE(24, 15).to(dev)
h = model.handle()
n = h.nparams
grads = torch.zeros(n + 1, dtype=torch.float32, device=dev)
m = torch.zeros(n, dtype=torch.float32, device=dev)
v = torch.zeros(n, dtype=torch.float32, device=dev)
for i in range(50):
    h.train_step(x[i * R:(i + 1) * R], model.flat, m, v, i + 1, 1e-3)
torch.cuda.synchronize()
buf = (ctypes.c_ulonglong * 64)()
L = native.lib()
L.bamd_debug_lat_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
rc = L.bamd_debug_lat_trace(buf, 64)
t = np.array(buf[:17], dtype=np.int64)
names = ["issue wf0/wf1", "x rows + bias", "L0", "L1", "L2", "L3", "L4", "L5", "L6", "L7+loss",
         "B7", "B6", "B5", "B4", "B3", "B2", "B1+loss sum"]
print("rc", rc, "total cycles", t[16] - t[0])
for i in range(1, 17):
    print(f"{names[i]:16s} {t[i] - t[i - 1]:7d} cycles")

t = np.array(buf[:64], dtype=np.int64)
for blk, o in ((0, 32), (150, 40)):
    d = t[o:o + 5]
    print(f"dw block {blk}: start +{d[0] - t[16]} after the chain's last stamp; map/state issued {d[1] - d[0]}, images+mfma {d[2] - d[1]}, "
          f"scatter idx + barrier {d[3] - d[2]}, adam + stores {d[4] - d[3]}")

# 4-row chain (lat4_chain_kernel, batches <= BALER_AMD_LAT4_ROWS): stamps 46..63 of workgroup 0, wave 0
t4 = t[46:64]
if t4[0] > 0:
    names4 = ["zero LDS + bias -> LDS + ring", "ring prologue", "x rows -> X_0", "L0", "L1", "L2", "L3", "L4", "L5", "L6", "L7+loss",
              "B7", "B6", "B5", "B4", "B3", "B2", "B1"]
    print("lat4 chain total cycles", t4[17] - t4[0])
    for i in range(1, 18):
        print(f"  {names4[i]:16s} {t4[i] - t4[i - 1]:7d} cycles")
