#!/usr/bin/env python3
"""Run-time-width classes against the exact instantiations and the layer-wise path (GPU box): python tools/bench_wide_class.py"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from baler_amd import native
from oracle import c_oracle as orc

os.environ["BALER_AMD_QUIET"] = "1"


def ms(fn, reps=3):
    fn(); torch.cuda.synchronize()
    for _ in range(8):
        fn()
    best = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record(); torch.cuda.synchronize()
        best.append(e0.elapsed_time(e1) / reps)
    return sorted(best)[2]


def handle(F, Z, env):
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        dims = orc.ae_dims(F, Z)
        h = native.Handle(dims, "fp32")
        p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
        h.load_params(p)
        return h, p
    finally:
        for k, v in old.items():
            os.environ.pop(k, None) if v is None else os.environ.__setitem__(k, v)


for (F, Z), rows in (((625, 7), 131072), ((2500, 25), 32768), ((900, 9), 131072), ((1024, 11), 131072), ((4096, 41), 32768), ((128, 13), 524288),
                     ((80, 16), 1000000), ((100, 1), 1000000)):
    x = torch.rand((rows, F), dtype=torch.float32, device="cuda")
    for tag, env in (("default", {}), ("class forced", {"BALER_AMD_WIDE_CLASS": "force"}), ("layer-wise", {"BALER_AMD_FORCE_GENERIC": "1"})):
        if tag == "class forced" and (F, Z) not in ((625, 7), (2500, 25)):
            continue
        h, p = handle(F, Z, env)
        z = h.encode(x)
        g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        st = {"t": 0}

        def steps():
            for i in range(50):
                st["t"] += 1
                h.train_step(x[i * 512:(i + 1) * 512], p, m, v, st["t"], 1e-3)
        te, td, tt, ts = ms(lambda: h.encode(x)), ms(lambda: h.decode(z)), ms(lambda: h.fwd_bwd(x, g), 2), ms(steps, 1) / 50
        print(f"AE({F},{Z}) {rows} rows [{tag:12s} path {h.path:11s}]: encode {te:7.3f} ms ({rows / te / 1e3:8.1f} M rows/s)  decode {td:7.3f} ms  "
              f"fwd_bwd {tt:8.3f} ms ({rows / tt / 1e3:7.1f} M rows/s)  512-row step {1e3 * ts:7.1f} us", flush=True)
        h.close()
    del x
