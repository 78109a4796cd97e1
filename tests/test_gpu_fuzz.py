"""Randomised shape coverage of the class instantiations (run-time widths), in the suite: seeded random (columns, latent) pairs
with ragged row counts through encode / decode / fwd_bwd / train_step against the scalar fp64 oracle (oracle/c_oracle) with the
suite's `rel()` (max of rel-L2 and max-norm).  models.py:122-139 builds AE(n_features, z_dim) for ANY table and baler.py:117-123
derives any latent; the hand-picked shapes of test_gpu_parity.py pin the class boundaries, these sweep inside them.

  fp32 / fp64: 40 shapes with F <= 79, Z <= 31 (narrow classes; BALER_AMD_LATENCY_ROWS = 512 at handle creation puts the second,
               larger batch of each fp32 shape on the throughput pair where the class has one); every other fp64 shape is drawn from
               F <= 127, Z <= 63 (the 4-row-chain classes of round 6: 64 .. 127 columns and / or a latent of 32 .. 63);
Rows are drawn clear of the LeakyReLU kink (test_gpu_parity.off_the_kink: a float32 pre-activation within 1e-6 of zero can carry
the other sign than its float64 twin -- a property of the comparison, not of a kernel).
"""
import os

import numpy as np
import pytest
import torch

from baler_amd import native
from oracle import c_oracle as orc
from test_gpu_parity import TOL32, TOL64, dev, make_handle, off_the_kink, rel

pytestmark = pytest.mark.gpu


def _adam_ref(dims, flat, x, lr=1e-3):
    """One optimiser step of the oracle from zero moments -> (loss, grads, new params)."""
    lo, go = orc.fwd_bwd(dims, flat, x)
    p, m, v = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
    orc.adam_step(p, go, m, v, 1, lr)
    return lo, go, p


def _check_shape(F, Z, mode, rng, sizes, seed):
    tol = TOL64 if mode == "fp64" else TOL32
    tdt = torch.float64 if mode == "fp64" else torch.float32
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 5000 + seed)
    h, p = make_handle(dims, flat, mode)
    worst = 0.0
    try:
        for n in sizes:
            x = off_the_kink(dims, flat, n, seed * 7 + n)
            xd = dev(x)
            z_ref = orc.encode(dims, flat, x)
            e = rel(h.encode(xd).cpu().numpy(), z_ref)
            assert e < tol, ("encode", F, Z, n, h.path, e)
            worst = max(worst, e)
            e = rel(h.decode(dev(z_ref, tdt)).cpu().numpy(), orc.decode(dims, flat, z_ref))
            assert e < tol, ("decode", F, Z, n, h.path, e)
            worst = max(worst, e)
            lo, go, pn = _adam_ref(dims, flat, x)
            g = torch.full_like(p, 3.0)
            h.fwd_bwd(xd, g)
            gh = g.cpu().numpy().astype(np.float64)
            e = rel(gh[:-1], go)
            assert e < tol and abs(gh[-1] - lo) < tol * abs(lo), ("fwd_bwd", F, Z, n, h.path, e, gh[-1], lo)
            worst = max(worst, e)
            # the one-call optimiser step from the same weights (fresh moments): parameters after the step
            p1, m1, v1 = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
            h.train_step(xd, p1, m1, v1, 1, 1e-3)
            # Adam's first step moves every parameter by lr * sign(g) (|g| >> eps): compare the MOVE, not the parameter -- an error
            # in the update would hide behind the 1e3 times larger weight otherwise; elements whose gradient is ~0 excepted
            live = np.abs(go) > 1e-6 * np.abs(go).max()
            mv, mv_ref = (p1.cpu().numpy().astype(np.float64)[:-1] - flat)[live], (pn - flat)[live]
            bar = 1e-3 if mode != "fp64" else 1e-9          # float32 parameters: the move is 1e-3 of the value, rounded at 6e-8 of it
            assert rel(mv, mv_ref) < bar, ("train_step", F, Z, n, h.path, rel(mv, mv_ref))
            h.load_params(p)                                 # back to the initial weights for the next size
    finally:
        h.close()
    return worst


@pytest.mark.parametrize("mode", ["fp32", "fp64"])
def test_class_fuzz(mode):
    rng = np.random.default_rng(20241008 if mode == "fp32" else 20241009)
    env = {"BALER_AMD_QUIET": "1"}
    if mode == "fp32":
        env["BALER_AMD_LATENCY_ROWS"] = "512"       # read at handle creation: batches above 512 rows take the throughput pair
    else:
        env["BALER_AMD_F64_REGCHAIN_BLKS"] = "48"   # read per call: from 768 rows on the per-wave register chain + tile blocks
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        worst, paths = 0.0, {}
        for k in range(40):
            F, Z = int(rng.integers(1, 80)), int(rng.integers(1, 32))
            if mode == "fp64" and k % 2:                # every other fp64 shape from the round-6 classes: up to 127 columns, a latent of up to 63
                F, Z = int(rng.integers(1, 128)), int(rng.integers(1, 64))
            sizes = (int(rng.integers(1, 500)), int(rng.integers(800, 1400)))
            worst = max(worst, _check_shape(F, Z, mode, rng, sizes, k))
        print(f"class fuzz {mode}: 40 shapes, worst {worst:.2e}")
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
