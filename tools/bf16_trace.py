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
buf = (ctypes.c_ulonglong * 1024)()
L = native.lib()
L.bamd_debug_bf16_trace.argtypes = [ctypes.c_void_p]
print("rc", L.bamd_debug_bf16_trace(buf))
T = np.array(buf[:], dtype=np.uint64).astype(np.int64).reshape(2, 4, 128)


def seg(part, label, ids):
    """ids: stamp ids in execution order, the first one = the phase's opening barrier; prints per-wave cycles since then"""
    base = T[part, :, ids[0]]
    cols = " | ".join(" ".join(f"{T[part, w, i] - base[w]:5d}" for w in range(4)) for i in ids[1:])
    print(f"  {label:26s} total {T[part, 0, ids[-1]] - base[0]:5d} | {cols}")


print("PART 0 (columns: per-wave cycles at [MFMAs done | epilogue issued | (dW done) | barrier passed])")
print("  iteration total", T[0, 0, 59] - T[0, 0, 0])
seg(0, "rows -> image 0", [0, 1])
seg(0, "fwd 0", [1, 40, 41, 2])
seg(0, "fwd 1", [2, 42, 43, 3])
seg(0, "fwd 2..5 register chain", [3, 54, 55])
seg(0, "fwd 6", [55, 52, 53, 8])
seg(0, "fwd 7 + loss", [8, 9])
seg(0, "bwd 7", [9, 84, 34, 85, 35])
seg(0, "bwd 6", [35, 82, 32, 83, 33])
seg(0, "bwd 5..2 register chain", [33, 56, 57])
seg(0, "dW 5..2", [57, 58, 59])
print("PART 1")
print("  iteration total", T[1, 0, 21] - T[1, 0, 0])
seg(1, "rows + dZ_1 -> images", [0, 1])
seg(1, "fwd 0", [1, 40, 41, 2])
seg(1, "bwd 1", [2, 72, 22, 73, 23])
seg(1, "bwd 0 (dW only)", [23, 20, 71, 21])
