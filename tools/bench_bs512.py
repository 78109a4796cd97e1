"""Strict reference batching (batch_size 512, one optimiser step per batch, sequential): us/step of the training loop.

    python tools/bench_bs512.py [--rows 1000000] [--bs 512] [--epochs 3]

Measures the same call sequence training.fit issues (bamd_fwd_bwd + bamd_adam_step per batch) and the single-call
form (bamd_train_step).  Prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baler_amd import native, synth                               # noqa: E402
from baler_amd.modules import models                              # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rows", type=int, default=1_000_000)
    ap.add_argument("--bs", type=int, default=512)
    ap.add_argument("--epochs", type=int, default=3)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    x64 = torch.from_numpy(synth.cms_rows(a.rows)).to(dev)
    feats = native.minmax(x64)
    x = native.normalize(x64, feats, out_dtype=torch.float64)
    model = models.AE(24, 15).to(dev)
    h = model.handle()
    flat = model.flat
    n = h.nparams
    grads = torch.zeros(n + 1, dtype=flat.dtype, device=dev)
    m = torch.zeros(n, dtype=flat.dtype, device=dev)
    v = torch.zeros(n, dtype=flat.dtype, device=dev)
    loss_acc = torch.zeros(1, dtype=torch.float64, device=dev)
    nb = (a.rows + a.bs - 1) // a.bs
    out = {"rows": a.rows, "batch_size": a.bs, "steps_per_epoch": nb}

    t = [0]
    def epoch_python():
        for i in range(nb):
            h.fwd_bwd(x[i * a.bs:(i + 1) * a.bs], grads)
            t[0] += 1
            h.adam_step(flat, grads, m, v, t[0], 1e-3, loss_accum=loss_acc)
    epoch_python()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.epochs):
        epoch_python()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.epochs
    out["python_loop_us_per_step"] = 1e6 * dt / nb
    out["python_loop_rows_per_s"] = a.rows / dt

    def epoch_fused():
        for i in range(nb):
            t[0] += 1
            h.train_step(x[i * a.bs:(i + 1) * a.bs], flat, m, v, t[0], 1e-3, loss_accum=loss_acc)
    epoch_fused()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.epochs):
        epoch_fused()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / a.epochs
    out["train_step_us_per_step"] = 1e6 * dt / nb
    out["train_step_rows_per_s"] = a.rows / dt
    print(json.dumps(out))


if __name__ == "__main__":
    main()
