#!/usr/bin/env python3
"""fp64 bamd_fwd_bwd of AE(24, 15) at large batches: the shipped weight-gradient tile blocks against the wave-owned-tile variants of the
round-6 experiment (profiles/r6_fp64_wave_owned_tiles.txt).  The variants and their switches (BALER_AMD_DW64Y_BLKS, BALER_AMD_DW64Y_RANGES,
BALER_AMD_DW64Y_LDS, BALER_AMD_DW64_CAP28_BLKS) exist only in the experiment commits e992b02 / c5eec6f (git checkout <commit> -- baler_amd/csrc,
make); against the shipped library every line measures the shipped path.  Gradients of the variants are compared with the first line's."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
from _gpu_warm import warm
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "fp64")
p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
h.load_params(p)
def ms(fn, reps):
    warm(60.0)
    for _ in range(3): fn()
    out = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): fn()
        e1.record(); torch.cuda.synchronize()
        out.append(e0.elapsed_time(e1) / reps)
    return sorted(out)[2]
for n in [int(a) for a in sys.argv[1:]] or [65536, 262144, 1000000]:
    x = torch.from_numpy(orc.normalize(synth.cms_rows(n))).cuda()
    ref = None
    for name, env in (("16-tile blocks, 32 ranges", {}), ("wave-owned tiles in registers, 64 ranges", {"BALER_AMD_DW64Y_BLKS": "1024", "BALER_AMD_DW64Y_RANGES": "64"}),
                      ("wave-owned tiles in registers, 128 ranges", {"BALER_AMD_DW64Y_BLKS": "1024", "BALER_AMD_DW64Y_RANGES": "128"}),
                      ("wave-owned tiles, slices via LDS, 64 ranges", {"BALER_AMD_DW64Y_BLKS": "1024", "BALER_AMD_DW64Y_LDS": "1"})):
        for k in ("BALER_AMD_DW64Y_BLKS", "BALER_AMD_DW64Y_RANGES", "BALER_AMD_DW64_CAP28_BLKS", "BALER_AMD_DW64Y_LDS"):
            os.environ.pop(k, None)
        os.environ.update(env)
        g = torch.zeros_like(p)
        h.fwd_bwd(x, g)
        torch.cuda.synchronize()
        if ref is None: ref = g.clone()
        err = float((g[:-1] - ref[:-1]).norm() / ref[:-1].norm())
        t = ms(lambda: h.fwd_bwd(x, g), max(3, min(20, 4000000 // n)))
        print(f"{n:8d} rows  {name:44s}: {t:7.3f} ms = {357000 * n / t / 1e9:6.2f} TFLOP/s ({357000 * n / t / 1e9 / 78.6:.3f})  rel. diff to the first {err:.1e}  loss {g[-1].item():.12g}", flush=True)
