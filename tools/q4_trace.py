"""Per-GEMM shader-clock timeline of chain64q_kernel's workgroup 0, wave 0 (needs a -DBAMD_Q4_TRACE build of fused64.hip):
    make -C baler_amd/csrc clean && make -C baler_amd/csrc -j8 HIPFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function -DBAMD_Q4_TRACE" LIB=../../.abl/q4trace.so
    (or two tools/abl_build.sh steps: fused64.hip and fused64q.hip both with -DBAMD_Q4_TRACE)
    BALER_AMD_LIB=$PWD/.abl/q4trace.so python tools/q4_trace.py [ROWS]      (on the GPU box)
"""
import ctypes, os, sys
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baler_amd import native, synth
from oracle import c_oracle as orc
R = int(sys.argv[1]) if len(sys.argv) > 1 else 512
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
x = torch.from_numpy(orc.normalize(synth.cms_rows(R * 20))).cuda()
m, v = torch.zeros_like(p), torch.zeros_like(p)
L = native.lib()
L.bamd_debug_q4_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
L.bamd_debug_dw64_trace.argtypes = [ctypes.c_void_p, ctypes.c_int]
names = ["start", "rows + biases + ring issued", "X_0 published", "L0 (13 groups, K 24)", "L1 (7, K 200)", "L2 (4, K 100)", "L3 (1, K 50, split 4)",
         "L4 (4, K 15)", "L5 (7, K 50)", "L6 (13, K 100)", "L7 + loss (2, K 200, split 2)", "B7 (13, K 24)", "B6 (7, K 200)", "B5 (4, K 100)",
         "B4 (1, K 50, split 4)", "B3 (4, K 15)", "B2 (7, K 50)", "B1 (13, K 100)", "loss partial"]
acc = np.zeros(19)
dw = np.zeros(5)
reps = 20
for i in range(60):
    k = i % 20
    h.train_step(x[k * R:(k + 1) * R], p, m, v, i + 1, 1e-3)
    if i >= 60 - reps:
        torch.cuda.synchronize()
        buf = (ctypes.c_ulonglong * 32)()
        L.bamd_debug_q4_trace(buf, 32)
        t = np.array(buf[:19], dtype=np.int64)
        acc += (t - t[0])
        dbuf = (ctypes.c_ulonglong * 8)()
        L.bamd_debug_dw64_trace(dbuf, 8)
        d = np.array(dbuf[:5], dtype=np.int64)
        dw += np.concatenate([[d[0] - t[18]], np.diff(d)])
acc /= reps
print(f"chain64q_kernel, {R} rows per step, workgroup 0 wave 0, mean of {reps} steps: total {acc[18]:.0f} cycles (s_memtime ticks)")
for i in range(1, 19):
    print(f"  {names[i]:34s} {acc[i] - acc[i - 1]:8.0f}")
dw /= reps
print("dw64_kernel<adam>, the workgroup of tile 150, thread 0 (cycles): starts %.0f after the chain's last stamp of workgroup 0;" % dw[0])
for nm, v in zip(("map entry, optimiser state and scatter indices requested -> known", "image slices + MFMAs", "LDS hand-over + barrier", "sum, Adam, stores"), dw[1:]):
    print(f"  {nm:66s} {v:8.0f}")
