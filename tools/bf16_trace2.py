"""Per-phase shader-clock timeline of the round-5 bf16 training pair (bf16_train2_kernel), workgroup 0, all four waves, LAST
iteration.  Needs a -DBAMD_BF16_TRACE build:  tools/abl_build.sh btrace bf16_train.hip -DBAMD_BF16_TRACE
    BALER_AMD_LIB=$PWD/.abl/btrace.so python tools/bf16_trace2.py          (on the GPU box)"""
import ctypes
import os
import sys

import numpy as np
import torch

os.environ["BALER_AMD_BF16_TRAIN_V2"] = "1"

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

P0 = ["top", "rows->regs", "fwd 0 (13)", "fwd 1 (49)", "fwd 2 (16)", "fwd 3 (2)", "fwd 4 (4)", "fwd 5 (14)", "fwd 6 (52)", "fwd 7 (14)",
      "loss + dZ_7", "barrier A", "bwd 7 (13)", "dW 7", "barrier B", "bwd 6 (49)", "bwd 5 (16)", "bwd 4 (2) + hand-off + next rows",
      "barrier D", "dW 6", "dW 5", "dW 4", "barrier E"]
P1 = ["top", "rows->regs + X_0", "fwd 0 (13)", "fwd 1 (49)", "fwd 2 (16)", "X_3, dZ_3 -> images", "barrier A", "bwd 3 (4)", "dW 3", "barrier B",
      "bwd 2 (14)", "bwd 1 (52) + next rows", "barrier D", "dW 2", "dW 1", "dW 0", "barrier E"]
for part, names in ((0, P0), (1, P1)):
    print(f"PART {part}: iteration total (wave 0) {T[part, 0, len(names) - 1] - T[part, 0, 0]} cycles; per phase, waves 0..3")
    for i in range(1, len(names)):
        d = T[part, :, i] - T[part, :, i - 1]
        print(f"  {names[i]:36s} " + " ".join(f"{int(v):6d}" for v in d))
