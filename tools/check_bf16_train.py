#!/usr/bin/env python3
"""bf16 training kernels vs the fp64 oracle: per-tensor gradient error at several batch sizes, then the rate at 1M rows.
  gpurun -- python tools/check_bf16_train.py"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from baler_amd import native, synth
from baler_amd.modules import models
from oracle import c_oracle as orc

dims = orc.ae_dims(24, 15)
flat = orc.formula_params(dims, 7)
raw = synth.cms_rows(20000)
x = orc.normalize(raw)
m = models.AE(24, 15, mode="bf16").load_flat(flat).to("cuda:0")
h = m.handle()
layout, _ = models.tensor_layout(dims)
worst = 0.0
for n in (64, 16, 272, 1000, 4096 + 17, 20000):
    xd = torch.as_tensor(x[:n]).cuda()
    g = torch.zeros_like(m.flat)
    h.fwd_bwd(xd, g)
    torch.cuda.synchronize()
    loss_ref, g_ref = orc.fwd_bwd(dims, flat, x[:n])
    gh = g.cpu().numpy().astype(np.float64)
    errs = []
    for key, off, shape in layout:
        k = int(np.prod(shape))
        errs.append(np.linalg.norm(gh[off:off + k] - g_ref[off:off + k]) / max(np.linalg.norm(g_ref[off:off + k]), 1e-300))
    tot = np.linalg.norm(gh[:-1] - g_ref) / np.linalg.norm(g_ref)
    worst = max(worst, tot)
    print(f"n={n:6d} loss {gh[-1]:.6f} ref {loss_ref:.6f} rel {abs(gh[-1] - loss_ref) / loss_ref:.2e}  grad rel-L2 {tot:.3e}  "
          f"per tensor max {max(errs):.3e} ({layout[int(np.argmax(errs))][0]})")
    g2 = torch.zeros_like(m.flat)
    h.fwd_bwd(xd, g2)
    assert torch.equal(g, g2), "not reproducible"
    # f32 input + normalise-on-load
    if n == 1000:
        feats = torch.as_tensor(np.stack([raw.min(0), raw.max(0) - raw.min(0)])).cuda()
        g3 = torch.zeros_like(m.flat)
        h.fwd_bwd(torch.as_tensor(raw[:n]).cuda(), g3, features=feats)
        print("   fused normalise: max |diff| vs pre-normalised", float((g3 - g).abs().max()), "of", float(g.abs().max()))
print("worst total rel-L2", worst)

rows = 1_000_000
xd = native.normalize(torch.as_tensor(synth.cms_rows(rows)).cuda(), native.minmax(torch.as_tensor(synth.cms_rows(rows)).cuda()))
g = torch.zeros_like(m.flat)
for _ in range(3):
    h.fwd_bwd(xd, g)
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(20):
    h.fwd_bwd(xd, g)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"bf16 fwd_bwd 1M rows: {dt * 1e3:.3f} ms = {rows / dt / 1e6:.1f} M rows/s = {357000 * rows / dt / 1e12:.1f} TFLOP/s")
mm, vv = torch.zeros_like(m.flat), torch.zeros_like(m.flat)
t0 = time.perf_counter()
for i in range(20):
    h.fwd_bwd(xd, g)
    h.adam_step(m.flat, g, mm, vv, i + 1, 1e-3)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / 20
print(f"bf16 train step 1M rows: {dt * 1e3:.3f} ms = {rows / dt / 1e6:.1f} M rows/s")
x512 = xd[:512 * 200]
t0 = time.perf_counter()
for i in range(200):
    h.train_step(x512[i * 512:(i + 1) * 512], m.flat, mm, vv, 21 + i, 1e-3)
torch.cuda.synchronize()
print(f"bf16 bs512 step: {(time.perf_counter() - t0) / 200 * 1e6:.1f} us")
