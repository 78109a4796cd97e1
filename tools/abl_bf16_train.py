#!/usr/bin/env python3
"""Interleaved A/B of bf16 training variants: python tools/abl_bf16_train.py lib1.so lib2.so ...  (each in its own process,
ROUNDS rounds interleaved; prints median / min of bamd_fwd_bwd at 1M rows per variant)."""
import os
import subprocess
import sys

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import os, sys, time
sys.path.insert(0, %r)
import torch
from baler_amd import native, synth
from baler_amd.modules import models
rows = int(os.environ.get("ABL_ROWS", "1000000"))
raw = torch.as_tensor(synth.cms_rows(rows)).cuda()
xd = native.normalize(raw, native.minmax(raw))
torch.manual_seed(0)
m = models.AE(24, 15, mode="bf16").to("cuda:0")
h = m.handle(); g = torch.zeros_like(m.flat)
for _ in range(5): h.fwd_bwd(xd, g)
torch.cuda.synchronize()
out = []
for r in range(int(os.environ.get("ABL_ROUNDS", "5"))):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20): h.fwd_bwd(xd, g)
    e1.record(); torch.cuda.synchronize()
    out.append(e0.elapsed_time(e1) / 20)
print("RES", " ".join("%%.4f" %% t for t in out), float(g.abs().sum()))
''' % R
libs = sys.argv[1:]
res = {l: [] for l in libs}
for rnd in range(int(os.environ.get("ABL_OUTER", "2"))):
    for l in libs:
        env = dict(os.environ, BALER_AMD_LIB=os.path.abspath(l))
        o = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True, timeout=600)
        line = [x for x in o.stdout.splitlines() if x.startswith("RES")]
        if not line:
            print(l, "FAILED", o.stderr[-500:]); continue
        v = line[0].split()[1:]
        res[l] += [float(x) for x in v[:-1]]
        chk = v[-1]
        print(f"{os.path.basename(l):24s} round {rnd}: " + " ".join(v[:-1]) + f"  checksum {chk}", flush=True)
for l in libs:
    if res[l]:
        r = sorted(res[l])
        print(f"{os.path.basename(l):24s} median {r[len(r)//2]:.4f} ms  min {r[0]:.4f} ms  -> {1e-3/r[len(r)//2]:.3f} G rows/s")
