"""Run-time-width classes for models other than the compiled-in shapes (reference models.py:122-139, 192-209 build
AE / CFD_dense_AE(n_features, z_dim) for ANY width, baler.py:117-123 derives any latent, and three shipped configs feed un-blocked
2-D data of whatever size the file has):

  64 .. 127 columns, latent <= 31   ImplInferClass<79|95|111|127, 31>: fused one-tile inference kernels + the small-batch training
                                    step (chain + weight-gradient tiles + fused Adam); large training batches layer-wise;
  48 .. 4096 columns, latent <= 63  ImplWide<4096, 15|31|63, true>: the wide-layer kernels (en1 / de4 streamed around the register
                                    chain, LDS-shared fragments) with the column count, the chunk count and the latent as kernel
                                    arguments: encode / decode / forward + loss and the two row-local launches of a training pass
                                    (the weight gradients are the layer-wise split-K kernels, as for the exact wide shapes).

Every call against the scalar fp64 oracle at ragged row counts with `rel()` (max of rel-L2 and max-norm, 1e-5), and the class
forced onto shapes that have an exact instantiation (BALER_AMD_WIDE_CLASS=force at bamd_create) against that instantiation.
"""
import os

import numpy as np
import pytest
import torch

from baler_amd import native
from oracle import c_oracle as orc
from test_gpu_parity import TOL32, dev, make_handle, off_the_kink, rel

pytestmark = pytest.mark.gpu


def _check(dims, flat, h, p, sizes, seed, per_tensor=True):
    F = dims[0]
    for n in sizes:
        x = off_the_kink(dims, flat, n, seed + n)
        z_ref = orc.encode(dims, flat, x)
        for xin in (dev(x), dev(x, torch.float32)):
            want = z_ref if xin.dtype == torch.float64 else orc.encode(dims, flat, x.astype(np.float32).astype(np.float64))
            assert rel(h.encode(xin, out_dtype=torch.float32).cpu().numpy(), want) < TOL32, ("encode", dims[0], dims[4], n, xin.dtype)
        rec_ref = orc.decode(dims, flat, z_ref)
        assert rel(h.decode(dev(z_ref, torch.float32)).cpu().numpy(), rec_ref) < TOL32, ("decode", F, n)
        assert rel(h.decode(dev(z_ref), out_dtype=torch.float64).cpu().numpy(), rec_ref) < TOL32, ("decode64", F, n)
        recon, loss = h.forward_loss(dev(x, torch.float32))
        fw = orc.forward(dims, flat, x.astype(np.float32).astype(np.float64))
        assert rel(recon.cpu().numpy(), fw) < TOL32, ("forward", F, n)
        assert abs(loss.item() - orc.loss(x.astype(np.float32).astype(np.float64), fw)) < TOL32 * loss.item()
        lo, go = orc.fwd_bwd(dims, flat, x.astype(np.float32).astype(np.float64))
        g = torch.full_like(p, 3.0)
        h.fwd_bwd(dev(x, torch.float32), g)
        gh = g.cpu().numpy().astype(np.float64)
        assert rel(gh[:-1], go) < TOL32 and abs(gh[-1] - lo) < TOL32 * lo, ("fwd_bwd", F, dims[4], n, rel(gh[:-1], go))
        if per_tensor:
            off = 0
            for l in range(8):
                for cnt in (dims[l + 1] * dims[l], dims[l + 1]):
                    assert rel(gh[off:off + cnt], go[off:off + cnt]) < 2 * TOL32, ("tensor", F, n, l, cnt)
                    off += cnt


@pytest.mark.parametrize("F,Z", [(80, 16), (100, 1), (127, 31), (96, 20)])
def test_mid_width_tables_fused_inference_and_small_batch_step(F, Z):
    """80 .. 127 columns: fused inference and the 512-row optimiser step (reference batch_size = 512, training.py:64-97) on the class
    kernels; the 512-row step must not fall back to the layer-wise path (a 16x cliff in round 4: 420 us vs 26 us).  Such a handle has
    two states: the wide class (inference, batches beyond the small-batch limit) and the small-batch class."""
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 300 + F)
    h, p = make_handle(dims, flat, "fp32")
    assert h.path == "fused"
    _check(dims, flat, h, p, (1, 17, 333, 513), F * 10)
    # the one-call step == fwd_bwd + adam_step bit for bit (both on the small-batch kernels)
    x = dev(off_the_kink(dims, flat, 512, 5))
    p1, p2 = p.clone(), p.clone()
    m1, v1, m2, v2 = (torch.zeros_like(p) for _ in range(4))
    h1, _ = make_handle(dims, flat, "fp32")
    h1.train_step(x, p1, m1, v1, 1, 1e-3)
    g = torch.zeros_like(p)
    h.fwd_bwd(x, g)
    h.adam_step(p2, g, m2, v2, 1, 1e-3)
    assert torch.equal(p1, p2) and torch.equal(m1, m2) and torch.equal(v1, v2)
    lo, go = orc.fwd_bwd(dims, flat, x.cpu().numpy())
    pn, mm, vv = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
    orc.adam_step(pn, go, mm, vv, 1, 1e-3)
    live = np.abs(go) > 1e-6 * np.abs(go).max()
    assert rel((p1.cpu().numpy().astype(np.float64)[:-1] - flat)[live], (pn - flat)[live]) < 1e-3


@pytest.mark.parametrize("F,Z", [(128, 13), (900, 9), (1024, 11), (4096, 41), (161, 2), (100, 40), (3999, 63)])
def test_wide_class_vs_oracle(F, Z):
    """ImplWide<4096, ZC, true>: any wide model with the reference's hidden widths (48 .. 4096 columns, a latent of up to 63)."""
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 400 + F)
    h, p = make_handle(dims, flat, "fp32")
    assert h.path == "fused"
    sizes = (1, 65, 300) if F <= 1024 else (1, 67)
    _check(dims, flat, h, p, sizes, F, per_tensor=F <= 1024)
    # normalise-on-load / un-normalise + truncation on store (float32 staging pass, as for the exact wide shapes)
    rng = np.random.default_rng(F)
    raw = rng.normal(size=(77, F)) * 5 + 2
    mn, rg = raw.min(0), raw.max(0) - raw.min(0)
    feats = dev(np.stack([mn, rg]))
    zn = orc.encode(dims, flat, (raw - mn) / rg)
    assert rel(h.encode(dev(raw), features=feats, out_dtype=torch.float32).cpu().numpy(), zn) < TOL32
    mask = np.zeros(F, dtype=np.uint8)
    mask[::5] = 1
    out = h.decode(dev(zn, torch.float32), features=feats, int_mask=dev(mask), out_dtype=torch.float64).cpu().numpy()
    pre = orc.decode(dims, flat, zn) * rg + mn
    m1 = mask == 1
    assert rel(out[:, ~m1], pre[:, ~m1]) < TOL32
    edge = np.abs(pre[:, m1] - np.round(pre[:, m1])) < 1e-4 * np.maximum(1.0, np.abs(pre[:, m1]))
    assert np.array_equal(out[:, m1][~edge], np.trunc(pre[:, m1])[~edge])


@pytest.mark.parametrize("F,Z,n", [(128, 13, 9001), (900, 9, 9001), (1024, 11, 9001), (4096, 41, 2100)])
def test_wide_class_throughput_launches_vs_oracle(F, Z, n, monkeypatch):
    """The class's THROUGHPUT launches (wide_train_fwd / bwd_kernel<..., WRT = true>, the split-K weight-gradient kernels, the
    two-tile encode) under an oracle assertion: batches above BALER_AMD_WIDE_SMALL_ROWS (8192) with default knobs -- the sizes
    `bench.py`'s wide_class numbers are measured at run the same launches -- and, for the 4096-column shape whose oracle pass costs
    9 MFLOP per row, the same launches forced onto a smaller batch (the limit is read per call).  Gradients per tensor."""
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 500 + F)
    h, p = make_handle(dims, flat, "fp32")
    assert h.path == "fused"
    if n <= 8192:
        monkeypatch.setenv("BALER_AMD_WIDE_SMALL_ROWS", "0")
    _check(dims, flat, h, p, (n,), F + 1)
    # one optimiser step from these weights: the move of every live parameter
    x = off_the_kink(dims, flat, n, F + 1 + n)
    lo, go = orc.fwd_bwd(dims, flat, x.astype(np.float32).astype(np.float64))
    pn, mm, vv = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
    orc.adam_step(pn, go, mm, vv, 1, 1e-3)
    p1, m1, v1 = p.clone(), torch.zeros_like(p), torch.zeros_like(p)
    h.train_step(dev(x, torch.float32), p1, m1, v1, 1, 1e-3)
    live = np.abs(go) > 1e-6 * np.abs(go).max()
    assert rel((p1.cpu().numpy().astype(np.float64)[:-1] - flat)[live], (pn - flat)[live]) < 1e-3


def test_wide_class_fuzz():
    """Seeded random wide-class shapes (128 .. 4096 columns, latent 1 .. 63) around the small-batch / throughput switch: with
    BALER_AMD_WIDE_SMALL_ROWS = 600 (read per call) each shape runs a ragged small batch, the last batch on the split launches (600
    rows) and the first on the throughput launches (601) through encode / decode / forward / fwd_bwd against the oracle."""
    rng = np.random.default_rng(20241010)
    old = {k: os.environ.get(k) for k in ("BALER_AMD_QUIET", "BALER_AMD_WIDE_SMALL_ROWS")}
    os.environ.update({"BALER_AMD_QUIET": "1", "BALER_AMD_WIDE_SMALL_ROWS": "600"})
    try:
        for k in range(8):
            F = int(rng.integers(128, 4097)) if k % 2 else int(rng.integers(128, 1200))
            Z = int(rng.integers(1, 64))
            dims = orc.ae_dims(F, Z)
            flat = orc.formula_params(dims, 7000 + k)
            h, p = make_handle(dims, flat, "fp32")
            assert h.path == "fused", (F, Z)
            sizes = (int(rng.integers(1, 500)), 600, 601) if F <= 1200 else (int(rng.integers(1, 200)), 601)
            try:
                _check(dims, flat, h, p, sizes, 100 * k, per_tensor=F <= 1200)
            finally:
                h.close()
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize("shape", [(625, 7), (2500, 25), (512, 6)])
def test_wide_class_forced_onto_exact_shapes(shape, monkeypatch):
    """The class kernels on the shapes that also have an exact instantiation: same results to float32 rounding (the class chunk that
    reaches beyond the row is a full tile with zero weights where the exact kernel runs a short r-major tile: different summation
    order in that one chunk), both within 1e-5 of the oracle."""
    dims = orc.ae_dims(*shape)
    flat = orc.formula_params(dims, 77)
    he, p = make_handle(dims, flat, "fp32")
    monkeypatch.setenv("BALER_AMD_WIDE_CLASS", "force")
    hc, _ = make_handle(dims, flat, "fp32")
    assert he.path == "fused" and hc.path == "fused"
    n = 200
    x = off_the_kink(dims, flat, n, 9)
    xd = dev(x, torch.float32)
    ze, zc = he.encode(xd, out_dtype=torch.float32), hc.encode(xd, out_dtype=torch.float32)
    z_ref = orc.encode(dims, flat, x.astype(np.float32).astype(np.float64))
    assert rel(zc.cpu().numpy(), z_ref) < TOL32 and rel(zc.cpu().numpy(), ze.cpu().numpy()) < TOL32
    de, dc = he.decode(ze), hc.decode(ze)
    assert rel(dc.cpu().numpy(), de.cpu().numpy()) < TOL32
    ge, gc = torch.zeros_like(p), torch.zeros_like(p)
    he.fwd_bwd(xd, ge)
    hc.fwd_bwd(xd, gc)
    lo, go = orc.fwd_bwd(dims, flat, x.astype(np.float32).astype(np.float64))
    assert rel(gc.cpu().numpy().astype(np.float64)[:-1], go) < TOL32 and rel(gc.cpu().numpy(), ge.cpu().numpy()) < TOL32


def test_wide_class_off_switch_and_bounds(monkeypatch):
    """BALER_AMD_WIDE_CLASS=0: such shapes run layer by layer (the independent cross-check); beyond 4096 columns or a latent above
    63 there is no class."""
    dims = orc.ae_dims(900, 9)
    flat = orc.formula_params(dims, 3)
    h, p = make_handle(dims, flat, "fp32")
    monkeypatch.setenv("BALER_AMD_WIDE_CLASS", "0")
    monkeypatch.setenv("BALER_AMD_QUIET", "1")
    hg, _ = make_handle(dims, flat, "fp32")
    assert h.path == "fused" and hg.path == "generic"
    x = dev(off_the_kink(dims, flat, 150, 4), torch.float32)
    assert rel(h.encode(x, out_dtype=torch.float32).cpu().numpy(), hg.encode(x, out_dtype=torch.float32).cpu().numpy()) < 2e-5
    g1, g2 = torch.zeros_like(p), torch.zeros_like(p)
    h.fwd_bwd(x, g1)
    hg.fwd_bwd(x, g2)
    assert rel(g1.cpu().numpy(), g2.cpu().numpy()) < 2e-5
    monkeypatch.delenv("BALER_AMD_WIDE_CLASS")
    for F, Z in ((4097, 10), (500, 64)):
        hh = native.Handle(orc.ae_dims(F, Z), "fp32")
        assert hh.path == "generic"
        hh.close()


@pytest.mark.parametrize("F,Z", [(80, 16), (64, 9)])
def test_mid_width_large_batches_run_chunked_on_the_small_batch_kernels(F, Z, monkeypatch):
    """Beyond the small-batch limit a 64 .. 127-column table trains chunk after chunk on the same two kernels (every chunk after the
    first adds to the gradient and the loss): several chunks (BALER_AMD_CLASS_CHUNK_ROWS) against the oracle, against ONE chunk, and
    against the layer-wise pass it replaced (BALER_AMD_CLASS_CHUNK_ROWS=0)."""
    monkeypatch.setenv("BALER_AMD_MID_HYBRID", "0")     # the small-batch class alone (by default the wide class takes the large batches)
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 11)
    n = 12288 + 4099                                   # beyond the default limit of 12288 rows, a ragged tail
    x = off_the_kink(dims, flat, n, 2)
    lo, go = orc.fwd_bwd(dims, flat, x)
    got = {}
    for tag, chunk in (("many", "4096"), ("one", "1048576"), ("layerwise", "0")):
        monkeypatch.setenv("BALER_AMD_CLASS_CHUNK_ROWS", chunk)
        h, p = make_handle(dims, flat, "fp32")
        g = torch.full_like(p, 7.0)
        h.fwd_bwd(dev(x), g)
        g2 = torch.zeros_like(p)
        h.fwd_bwd(dev(x), g2)
        assert torch.equal(g, g2)                      # a fixed order whatever the chunking
        got[tag] = g.cpu().numpy().astype(np.float64)
        h.close()
    for tag, gh in got.items():
        assert rel(gh[:-1], go) < 2 * TOL32 and abs(gh[-1] - lo) < TOL32 * lo, (tag, rel(gh[:-1], go))
    assert rel(got["many"], got["one"]) < TOL32 and rel(got["many"], got["layerwise"]) < 2 * TOL32


@pytest.mark.parametrize("F,Z", [(80, 16), (64, 9), (127, 31), (100, 1)])
def test_mid_width_two_state_handle_follows_the_optimiser(F, Z, monkeypatch):
    """A 64 .. 127-column handle keeps two packed copies of its weights (small-batch class + wide class).  Whatever the order of small
    steps, large passes and inference calls, every call must see the CURRENT parameters: compared with a handle of the small-batch
    class alone (BALER_AMD_MID_HYBRID=0, one state) driven through the same sequence, and with the oracle."""
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 21 + F)
    xs = dev(off_the_kink(dims, flat, 512, 3))
    xl = dev(off_the_kink(dims, flat, 12288 + 1040, 4))

    def drive():
        h, p = make_handle(dims, flat, "fp32")
        m, v = torch.zeros_like(p), torch.zeros_like(p)
        out = []
        h.train_step(xs, p, m, v, 1, 1e-3)                      # small: the class kernels, Adam fused
        out.append(h.encode(xl[:777]).cpu().numpy())            # inference right after a fused small step
        g = torch.zeros_like(p)
        h.fwd_bwd(xl, g)                                        # large pass on the parameters of step 1
        out.append(g.cpu().numpy().astype(np.float64))
        h.adam_step(p, g, m, v, 2, 1e-3)
        g2 = torch.zeros_like(p)
        h.fwd_bwd(xs, g2)                                       # small pass on the parameters of step 2
        out.append(g2.cpu().numpy().astype(np.float64))
        rec, loss = h.forward_loss(xl[:300])
        out.append(rec.cpu().numpy())
        h.train_step(xl, p, m, v, 3, 1e-3)                      # a large one-call step (fwd_bwd + Adam inside the library)
        out.append(h.decode(h.encode(xs)).cpu().numpy())
        out.append(p.cpu().numpy().astype(np.float64))
        path = h.path
        h.close()
        return path, out

    path2, two = drive()
    monkeypatch.setenv("BALER_AMD_MID_HYBRID", "0")
    path1, one = drive()
    assert path2 == "fused" and path1 == "fused-infer"
    for k, (a, b) in enumerate(zip(two, one)):
        assert rel(a, b) < 2 * TOL32, (k, rel(a, b))
    # ... and the oracle on the same sequence (first three products)
    st = orc.FitState(dims, flat)
    _, g0 = orc.fwd_bwd(dims, st.params, xs.cpu().numpy())
    orc.adam_step(st.params, g0, st.m, st.v, 1, 1e-3)
    assert rel(two[0], orc.encode(dims, st.params, xl[:777].cpu().numpy())) < 2 * TOL32
    lo, g1 = orc.fwd_bwd(dims, st.params, xl.cpu().numpy())
    assert rel(two[1][:-1], g1) < 2 * TOL32 and abs(two[1][-1] - lo) < TOL32 * lo


@pytest.mark.parametrize("shape", [(2500, 25), (625, 7), (900, 9), (100, 40)])
def test_wide_models_small_training_batches(shape, monkeypatch):
    """The reference trains its wide models with SMALL batches (CFD_project_still: batch_size = 60, exafel 1 .. 36, hurricane_isabel 85,
    CFD_project_animation 6000).  Up to 8192 rows the three wide products of a row group are split over workgroups (wide_small_in_kernel:
    en1 and de4's input-gradient product over the wide dimension, partial sums added in split order; wide_small_out_kernel: de4 over its
    output tiles): against the oracle, against the one-launch kernels (BALER_AMD_WIDE_SMALL_ROWS=0 is read once per process: compared
    through the oracle instead), bitwise repeatable, and a training step through it."""
    F, Z = shape
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 900 + F)
    h, p = make_handle(dims, flat, "fp32")
    for n in (1, 60, 85, 700, 6000):
        if F > 1000 and n > 1000:
            continue
        x = off_the_kink(dims, flat, n, n).astype(np.float32)
        lo, go = orc.fwd_bwd(dims, flat, x.astype(np.float64))
        g, g2 = torch.full_like(p, 5.0), torch.zeros_like(p)
        h.fwd_bwd(dev(x, torch.float32), g)
        h.fwd_bwd(dev(x, torch.float32), g2)
        assert torch.equal(g, g2)
        gh = g.cpu().numpy().astype(np.float64)
        assert rel(gh[:-1], go) < 2 * TOL32 and abs(gh[-1] - lo) < TOL32 * lo, (F, n, rel(gh[:-1], go))
        # the validation pass of the same batch (training.py:104-137) on the split launches: reconstruction (float32 / float64) and loss
        want = orc.forward(dims, flat, x.astype(np.float64))
        for od in (torch.float32, torch.float64):      # (the reconstruction has the rows' type)
            recon, loss = h.forward_loss(dev(x, od))
            assert recon.dtype == od and rel(recon.cpu().numpy(), want) < TOL32, (F, n, od)
            assert abs(loss.item() - ((want - x) ** 2).sum() / F) < TOL32 * max(1.0, ((want - x) ** 2).sum() / F)
    x = dev(off_the_kink(dims, flat, 60, 7), torch.float32)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    # the one-call step (Adam inside the weight-gradient launch) == fwd_bwd + adam_step, bit for bit, over two steps, loss sum included
    h2, p2 = make_handle(dims, flat, "fp32")
    m2, v2, g2 = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    la, lb = torch.zeros(1, dtype=torch.float64, device="cuda"), torch.zeros(1, dtype=torch.float64, device="cuda")
    pa, ma, va = p.clone(), m.clone(), v.clone()
    for t in (1, 2):
        h.train_step(x, pa, ma, va, t, 1e-3, loss_accum=la)
        h2.fwd_bwd(x, g2)
        h2.adam_step(p2, g2, m2, v2, t, 1e-3, loss_accum=lb)
    assert torch.equal(pa, p2) and torch.equal(ma, m2) and torch.equal(va, v2) and torch.equal(la, lb)
    assert rel(h.encode(x, out_dtype=torch.float32).cpu().numpy(), h2.encode(x, out_dtype=torch.float32).cpu().numpy()) == 0.0
    h2.close()
    h.load_params(p)
    h.train_step(x, p, m, v, 1, 1e-3)
    lo, go = orc.fwd_bwd(dims, flat, x.cpu().numpy().astype(np.float64))
    pn, mm, vv = flat.copy(), np.zeros_like(flat), np.zeros_like(flat)
    orc.adam_step(pn, go, mm, vv, 1, 1e-3)
    live = np.abs(go) > 1e-6 * np.abs(go).max()
    assert rel((p.cpu().numpy().astype(np.float64)[:-1] - flat)[live], (pn - flat)[live]) < 1e-3
    assert rel(h.encode(x, out_dtype=torch.float32).cpu().numpy(), orc.encode(dims, pn, x.cpu().numpy().astype(np.float64))) < 1e-4


@pytest.mark.parametrize("shape,n", [((2500, 25), 60), ((625, 7), 1), ((625, 7), 36), ((40, 40), 512), ((900, 9), 85), ((100, 70), 300)])
def test_layerwise_path_at_the_references_batch_sizes_fp64(shape, n):
    """The reference's own arithmetic (float64) at its own batch sizes on the models without fp64 fused kernels: the layer-wise path on the
    one-tile-per-workgroup kernels (gemm_small_k for the forward / loss / input-gradient products, dw_small_all_k for every weight gradient)
    against the oracle at the float64 bar; two optimiser steps follow the oracle's."""
    F, Z = shape
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, 50 + F)
    h = native.Handle(dims, "fp64")
    p = torch.from_numpy(np.concatenate([flat, [0.0]])).cuda()
    h.load_params(p)
    x = np.random.default_rng(n).random((n, F))
    lo, go = orc.fwd_bwd(dims, flat, x)
    g = torch.full_like(p, 3.0)
    h.fwd_bwd(dev(x), g)
    gh = g.cpu().numpy()
    assert rel(gh[:-1], go) < 1e-11 and abs(gh[-1] - lo) < 1e-11 * lo
    assert rel(h.encode(dev(x)).cpu().numpy(), orc.encode(dims, flat, x)) < 1e-11
    rec, loss = h.forward_loss(dev(x))
    assert rel(rec.cpu().numpy(), orc.forward(dims, flat, x)) < 1e-11
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    st = orc.FitState(dims, flat)
    for t in (1, 2):
        h.train_step(dev(x), p, m, v, t, 1e-3)
        _, gt = orc.fwd_bwd(dims, st.params, x)
        orc.adam_step(st.params, gt, st.m, st.v, t, 1e-3)
    assert rel(p.cpu().numpy()[:-1], st.params) < 1e-9
    h.close()
