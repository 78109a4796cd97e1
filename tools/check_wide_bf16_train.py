#!/usr/bin/env python3
"""bf16 training of the wide models (BAMD_MODE_BF16 handles of CFD_dense_AE(2500, 25) / (625, 7) / the 512-column model): gradients of one
pass against the fp64 oracle at ragged sizes, and the rate at 32,768 frames against the float32 launches (BALER_AMD_BF16_WIDE_TRAIN=0):
   gpurun -- python tools/check_wide_bf16_train.py"""
import os, subprocess, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
if len(sys.argv) > 1 and sys.argv[1] == "rate":
    import numpy as np, torch
    from baler_amd import native
    from oracle import c_oracle as orc
    for shape, n in (((2500, 25), 32768), ((625, 7), 131072)):
        dims = orc.ae_dims(*shape)
        h = native.Handle(dims, "bf16")
        p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
        h.load_params(p)
        x = torch.rand((n, shape[0]), dtype=torch.float32, device="cuda")
        g = torch.zeros_like(p)
        h.fwd_bwd(x, g); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): h.fwd_bwd(x, g)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 5
        print(f"RES {shape} {n} frames: fwd_bwd {ms:.3f} ms = {n / ms / 1e3:.1f} M frames/s, loss {float(g[-1]):.6f}")
    sys.exit(0)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
worst = 0.0
for shape, sizes in (((2500, 25), (1, 33, 300, 1037)), ((625, 7), (17, 129, 1000)), ((512, 6), (65, 1000))):
    dims = orc.ae_dims(*shape)
    flat = orc.formula_params(dims, 23)
    h = native.Handle(dims, "bf16")
    p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    for n in sizes:
        x = np.random.default_rng(n).random((n, shape[0]))
        lo, go = orc.fwd_bwd(dims, flat, x)
        g = torch.full_like(p, 7.0)
        h.fwd_bwd(torch.from_numpy(x).float().cuda(), g)
        gh = g.cpu().numpy().astype(np.float64)
        tot = np.linalg.norm(gh[:-1] - go) / np.linalg.norm(go)
        per, off = [], 0
        for l in range(8):
            for cnt in (dims[l] * dims[l + 1], dims[l + 1]):
                a, b = gh[off:off + cnt], go[off:off + cnt]
                per.append(np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-30)); off += cnt
        worst = max(worst, tot)
        print(f"{shape} n={n:5d} loss {gh[-1]:.6f} ref {lo:.6f} rel {abs(gh[-1] - lo) / lo:.2e}  grad rel-L2 {tot:.3e}  per tensor max {max(per):.3e} (tensor {int(np.argmax(per))})")
print("worst total rel-L2", worst)
for name, env in (("bf16 wide products", {}), ("float32 launches", {"BALER_AMD_BF16_WIDE_TRAIN": "0"})):
    o = subprocess.run([sys.executable, __file__, "rate"], env=dict(os.environ, **env), capture_output=True, text=True)
    print(name, [l for l in o.stdout.splitlines() if l.startswith("RES")] or o.stderr[-500:])
