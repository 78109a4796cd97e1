"""GPU parity on the edge cases of the normalisation the reference does not guard (data_processing.py:113-153): a constant
column (max == min -> 0/0), NaN cells (np.min / np.max propagate them) and +-inf cells, through bamd_minmax, bamd_col_minmax,
bamd_normalize, bamd_renormalize (with the int mask) and the fused normalise-on-load of bamd_encode / bamd_forward_loss /
bamd_fwd_bwd.  Expected values: tests/golden/g16_edge.npz (generated from the imported reference, tools/gen_golden_edge.py)
and the CPU oracle on the same inputs.  Bar: the NaN / inf PATTERN identical, every finite value bit-exact for the fp64
element-wise work and within the mode's tolerance for the model."""
import numpy as np
import pytest
import torch

from baler_amd import native, synth
from oracle import c_oracle as orc

pytestmark = pytest.mark.gpu


def dev(a, dtype=None):
    t = torch.as_tensor(np.ascontiguousarray(a))
    if dtype is not None:
        t = t.to(dtype)
    return t.cuda().contiguous()


def same_with_nans(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def rel(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_minmax_normalize_renormalize_edge_golden(golden, tag):
    g = golden("g16_edge.npz")
    raw, feats = g["raw_" + tag], g["features_" + tag]
    d = dev(raw)
    got = native.minmax(d).cpu().numpy()
    assert same_with_nans(got, feats)
    mm = native.col_minmax(d).cpu().numpy()
    with np.errstate(all="ignore"):
        assert same_with_nans(mm[0], feats[0]) and same_with_nans(mm[1] - mm[0], feats[1])
    normed = native.normalize(d, dev(feats), torch.float64).cpu().numpy()
    assert same_with_nans(normed, g["normalized_" + tag])
    ren = native.renormalize(dev(g["normalized_" + tag]), dev(feats)).cpu().numpy()
    assert same_with_nans(ren, g["renormalized_" + tag])
    # float32 output of the same normalisation: one rounding of the float64 quotient
    n32 = native.normalize(d, dev(feats), torch.float32).cpu().numpy()
    with np.errstate(all="ignore"):
        assert same_with_nans(n32, g["normalized_" + tag].astype(np.float32))


@pytest.mark.parametrize("n,where", [(1, 0), (17, 16), (5000, 0), (5000, 4999), (200_003, 123_457)])
@pytest.mark.parametrize("dtype", [torch.float64, torch.float32])
def test_minmax_nan_and_inf_anywhere(n, where, dtype):
    """A NaN cell poisons its column's min / max / range wherever it sits in the table (first row, last row, inside any
    workgroup's share of a large table); +-inf cells give the extrema numpy gives.  Other columns stay bit-exact."""
    x = synth.cms_rows(n)
    if dtype == torch.float32:
        x = x.astype(np.float32).astype(np.float64)
    x[where, 2] = np.nan
    x[where, 7] = np.inf
    x[n - 1 - where, 11] = -np.inf
    with np.errstate(all="ignore"):
        want = orc.find_minmax(x)
        assert same_with_nans(want, np.array([x.min(0), x.max(0) - x.min(0)]))       # the oracle is numpy here
        got = native.minmax(dev(x, dtype)).cpu().numpy()
        assert same_with_nans(got, want)
        mm = native.col_minmax(dev(x, dtype)).cpu().numpy()
        assert same_with_nans(mm, np.array([x.min(0), x.max(0)]))
    assert np.isnan(got[:, 2]).all() and np.isneginf(got[0, 11])
    assert np.isposinf(got[1, 7]) if n > 1 else np.isnan(got[1, 7])      # one row: range = inf - inf


def test_renormalize_int_mask_edge():
    """x * range + min with truncation toward zero on the int columns (baler.py:426-435): negative values, values just
    below an integer, NaN / inf in NON-int columns.  (astype(int) of a NaN / inf is undefined behaviour in numpy -- INT64_MIN on
    x86 -- so int columns are only exercised with finite values; the kernel leaves a NaN there a NaN.)"""
    rng = np.random.default_rng(5)
    normed = rng.uniform(-0.2, 1.2, size=(1003, 24))
    normed[5, 1] = np.nan
    normed[6, 2] = np.inf
    mins = rng.uniform(-50, 50, size=24)
    rngs = rng.uniform(0.5, 200, size=24)
    rngs[4] = 0.0                                    # constant column: renormalises to its minimum
    mask = np.zeros(24, dtype=np.uint8)
    mask[[0, 4, 9, 23]] = 1
    with np.errstate(all="ignore"):
        want = orc.cast_int_cols(orc.renormalize(normed, mins, rngs), mask)
    got = native.renormalize(dev(normed), dev(np.array([mins, rngs])), dev(mask)).cpu().numpy()
    assert same_with_nans(got, want)
    assert np.array_equal(got[:, 0], np.trunc(normed[:, 0] * rngs[0] + mins[0])) and (got[:, 4] == np.trunc(mins[4])).all()


@pytest.mark.parametrize("mode,tol", [("fp32", 1e-5), ("fp64", 1e-11), ("bf16", 2e-2)])
def test_fused_normalise_on_load_one_poisoned_row(golden, mode, tol):
    """Table A has one +inf cell: its row normalises to a NaN, every other row is finite.  bamd_encode with the features
    fused into the load must give NaN for exactly that row and the reference's latent for the rest."""
    g = golden("g16_edge.npz")
    dims = orc.ae_dims(24, 15)
    flat = orc.formula_params(dims, int(g["seed"]))
    h = native.Handle(dims, mode)
    h.load_params(dev(np.concatenate([flat, [0.0]]), torch.float64 if mode == "fp64" else torch.float32))
    feats = dev(g["features_a"])
    z_ref = g["z_a"]
    bad = np.isnan(z_ref).any(axis=1)
    for dt in (torch.float64, torch.float32):
        z = h.encode(dev(g["raw_a"], dt), features=feats).cpu().numpy().astype(np.float64)
        assert np.array_equal(np.isnan(z), np.isnan(z_ref)), (mode, dt)
        # float32 rows lose the low bits of the raw values before the fp64 normalisation: compare at 1e-5 at best
        assert rel(z[~bad], z_ref[~bad]) < (tol if dt == torch.float64 else max(tol, 1e-5)), (mode, dt)
    # the same through the pre-normalised table
    z2 = h.encode(dev(g["normalized_a"])).cpu().numpy().astype(np.float64)
    assert np.array_equal(np.isnan(z2), np.isnan(z_ref)) and rel(z2[~bad], z_ref[~bad]) < tol
    # forward + loss: the poisoned row makes the loss a NaN, as in the oracle
    recon, loss = h.forward_loss(dev(g["raw_a"]), features=feats)
    r = recon.cpu().numpy()
    assert np.isnan(loss.item()) and np.isnan(r[11]).all() and np.isfinite(np.delete(r, 11, axis=0)).all()
    want = orc.forward(dims, flat, g["normalized_a"])
    assert rel(np.delete(r, 11, axis=0), np.delete(want, 11, axis=0)) < tol


@pytest.mark.parametrize("mode", ["fp32", "fp64", "bf16"])
@pytest.mark.parametrize("n", [41, 4096 + 41])
def test_fwd_bwd_nan_rows_poison_loss_and_gradients(golden, mode, n):
    """One NaN row in a batch: loss and every gradient the row reaches are NaN in the oracle (and in torch); the HIP
    kernels must not drop it (a min/max-style select or a v_med3 would) on either training path (small-batch kernels at
    41 rows, the throughput pair / bf16 kernels at 4137)."""
    g = golden("g16_edge.npz")
    dims = orc.ae_dims(24, 15)
    flat = orc.formula_params(dims, int(g["seed"]))
    raw = g["raw_a"]
    if n > raw.shape[0]:
        raw = np.concatenate([synth.cms_rows(n - raw.shape[0], row0=7000), raw])       # the poisoned row sits in the last tile
    feats_np = g["features_a"]
    with np.errstate(all="ignore"):
        normed = (raw - feats_np[0]) / feats_np[1]
        loss_ref, g_ref = orc.fwd_bwd(dims, flat, normed)
    assert np.isnan(loss_ref) and np.isnan(g_ref).all()
    pdt = torch.float64 if mode == "fp64" else torch.float32
    h = native.Handle(dims, mode)
    h.load_params(dev(np.concatenate([flat, [0.0]]), pdt))
    grads = torch.zeros(len(flat) + 1, dtype=pdt, device="cuda")
    h.fwd_bwd(dev(raw), grads, features=dev(feats_np))
    got = grads.cpu().numpy()
    assert np.isnan(got[-1]), "loss"
    assert np.isnan(got[:-1]).all(), f"{int(np.isfinite(got[:-1]).sum())} finite gradient entries"


def test_decode_fused_renormalise_nan_latent():
    """decode of a NaN latent row: NaN reconstruction for that row only, also through the fused un-normalise + int truncation."""
    dims = orc.ae_dims(24, 15)
    flat = orc.formula_params(dims, 3)
    h = native.Handle(dims, "fp32")
    h.load_params(dev(np.concatenate([flat, [0.0]]), torch.float32))
    z = np.random.default_rng(2).normal(size=(70, 15))
    z[33, 4] = np.nan
    feats = np.array([np.linspace(-3, 3, 24), np.linspace(1, 9, 24)])
    mask = np.zeros(24, dtype=np.uint8)
    mask[[1, 2]] = 1
    out = h.decode(dev(z), features=dev(feats), int_mask=dev(mask)).cpu().numpy()
    with np.errstate(all="ignore"):
        want = orc.cast_int_cols(orc.renormalize(orc.decode(dims, flat, z), feats[0], feats[1]), mask)
    ok = np.ones(70, dtype=bool)
    ok[33] = False
    assert np.isnan(out[33]).all() and np.isfinite(out[ok]).all()
    # truncation flips an integer when the fp32 pre-image sits within 1e-5 of it: compare the non-int columns numerically
    cols = mask == 0
    assert rel(out[ok][:, cols], want[ok][:, cols]) < 1e-5
    assert np.mean(out[ok][:, ~cols] == want[ok][:, ~cols]) > 0.99


@pytest.mark.parametrize("shape", [(2500, 25), (512, 6)])
def test_wide_bf16_loader_kernels_keep_a_nan_in_its_row(shape):
    """The bf16 encode of the wide models streams rows through an LDS ring shared by four compute waves and two loader waves, the
    decode stores 256-byte windows assembled from an 8-tile ring: a NaN / inf cell must poison exactly its own row (latent /
    reconstruction), whichever ring slot, row tile or window it travels through, and every other row must equal the clean run's."""
    F, Z = shape
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 3)
    h = native.Handle(dims, "bf16")
    h.load_params(dev(np.concatenate([flat, [0.0]]), torch.float32))
    n = 777                                              # six full 128-row groups + a ragged one
    rng = np.random.default_rng(9)
    x = rng.random((n, F)).astype(np.float32)
    clean = h.encode(dev(x), out_dtype=torch.float32).cpu().numpy()
    bad_rows = [0, 127, 128, 300, 776]
    xb = x.copy()
    for k, r in enumerate(bad_rows):
        xb[r, (37 * k + 5) % F] = np.nan if k % 2 == 0 else np.inf
    z = h.encode(dev(xb), out_dtype=torch.float32).cpu().numpy()
    ok = np.ones(n, dtype=bool)
    ok[bad_rows] = False
    assert not np.isfinite(z[bad_rows]).all(axis=1).any(), "a poisoned row came out finite"
    assert np.isfinite(z[ok]).all() and np.array_equal(z[ok], clean[ok])
    zc = rng.normal(size=(n, Z)).astype(np.float32)
    dclean = h.decode(dev(zc)).cpu().numpy()
    zb = zc.copy()
    for r in bad_rows:
        zb[r, r % Z] = np.nan
    d = h.decode(dev(zb)).cpu().numpy()
    assert np.isnan(d[bad_rows]).all(), "a NaN latent must reach every column of its row"
    assert np.isfinite(d[ok]).all() and np.array_equal(d[ok], dclean[ok])
