"""bf16 MFMA training (BAMD_MODE_BF16: bamd_fwd_bwd / bamd_train_step on v_mfma_f32_16x16x32_bf16, fp32 master weights and
fp32 Adam).  A THROUGHPUT mode with its own acceptance bar (the 1e-5 parity mode is fp32):
  one-step gradients  rel-L2 <= 2e-2 vs the fp64 oracle (whole vector; <= 5e-2 per tensor), loss <= 2e-3
  a whole CLI run     loss curve of the C1 workload within 5 % of the reference's (fixture g7)
Bias and weights are rounded to bfloat16 once per step from the fp32 master copy; accumulation is fp32; reductions over the
batch run in a fixed order, so results are bitwise reproducible."""
import numpy as np
import pytest
import torch

from baler_amd import native, synth
from oracle import c_oracle as orc

pytestmark = pytest.mark.gpu

DIMS = orc.ae_dims(24, 15)


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def handle(flat, mode="bf16", dims=DIMS):
    h = native.Handle(dims, mode)
    p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    return h, p


@pytest.fixture(autouse=True)
def _bf16_kernels_for_every_batch(monkeypatch):
    """BF16 handles train batches of <= 3072 rows on the fp32 small-batch kernels (faster there); these tests are about the bf16
    kernels, at every size."""
    monkeypatch.setenv("BALER_AMD_BF16_SMALL_ROWS", "0")


@pytest.fixture(scope="module")
def data():
    raw = synth.cms_rows(20000)
    return raw, orc.normalize(raw)


@pytest.mark.parametrize("n", [1, 16, 63, 64, 65, 272, 1000, 4113, 20000])
def test_gradients_vs_oracle(n, data):
    raw, x = data
    flat = orc.formula_params(DIMS, 7)
    h, p = handle(flat)
    g = torch.zeros_like(p)
    h.fwd_bwd(torch.as_tensor(x[:n]).cuda(), g)
    loss_ref, g_ref = orc.fwd_bwd(DIMS, flat, x[:n])
    gh = g.cpu().numpy().astype(np.float64)
    assert abs(gh[-1] - loss_ref) < 2e-3 * loss_ref
    if n >= 64:
        assert rel(gh[:-1], g_ref) < 2e-2
    if n >= 272:          # per tensor (a few rows give en1 a gradient that is mostly rounding noise)
        off = 0
        for l in range(8):
            for k in (DIMS[l + 1] * DIMS[l], DIMS[l + 1]):
                assert rel(gh[off:off + k], g_ref[off:off + k]) < 5e-2, (l, k)
                off += k
    g2 = torch.zeros_like(p)
    h.fwd_bwd(torch.as_tensor(x[:n]).cuda(), g2)
    assert torch.equal(g, g2)                                   # fixed-order reductions
    if n in (65, 1000):
        # float32 rows, and raw rows normalised on load (float64 arithmetic, one rounding): the same bits
        g3 = torch.zeros_like(p)
        feats = torch.as_tensor(np.stack([raw.min(0), raw.max(0) - raw.min(0)])).cuda()
        h.fwd_bwd(torch.as_tensor(raw[:n]).cuda(), g3, features=feats)
        assert torch.equal(g3, g)
        g4 = torch.zeros_like(p)
        h.fwd_bwd(torch.as_tensor(x[:n].astype(np.float32)).cuda(), g4)
        assert rel(g4.cpu().numpy(), gh) < 1e-3


@pytest.mark.parametrize("z", [12, 8, 6])
def test_other_latent_sizes(z, data):
    _, x = data
    dims = orc.ae_dims(24, z)
    flat = orc.formula_params(dims, 3)
    h, p = handle(flat, dims=dims)
    g = torch.zeros_like(p)
    h.fwd_bwd(torch.as_tensor(x[:700]).cuda(), g)
    loss_ref, g_ref = orc.fwd_bwd(dims, flat, x[:700])
    gh = g.cpu().numpy().astype(np.float64)
    assert rel(gh[:-1], g_ref) < 2e-2 and abs(gh[-1] - loss_ref) < 2e-3 * loss_ref


def test_training_steps_track_the_fp32_run(data):
    """40 Adam steps at batch 512: the bf16 run's losses stay within 2 % of the fp32 parity mode's, step by step, and
    bamd_train_step == bamd_fwd_bwd + bamd_adam_step bit for bit."""
    _, x = data
    flat = orc.formula_params(DIMS, 11)
    runs = {}
    for mode in ("fp32", "bf16", "bf16-one-call"):
        h, p = handle(flat, "fp32" if mode == "fp32" else "bf16")
        m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        losses = []
        for t in range(1, 41):
            xb = torch.as_tensor(x[(t - 1) * 512 % 19000:][:512]).cuda()
            if mode == "bf16-one-call":
                h.train_step(xb, p, m, v, t, 1e-3, grads=g)
            else:
                h.fwd_bwd(xb, g)
                h.adam_step(p, g, m, v, t, 1e-3)
            losses.append(float(g[-1]))
        runs[mode] = (np.array(losses), p.clone(), h)
    assert np.max(np.abs(runs["bf16"][0] / runs["fp32"][0] - 1)) < 2e-2
    assert runs["bf16"][0][-1] < 0.5 * runs["bf16"][0][0]                      # it trains
    assert torch.equal(runs["bf16"][1], runs["bf16-one-call"][1])
    # inference on the SAME handle sees the updated weights (the inference fragments are re-rounded lazily)
    h, p = runs["bf16"][2], runs["bf16"][1]
    xe = torch.as_tensor(x[:1000]).cuda()
    z = h.encode(xe)
    fresh, _ = handle(p.cpu().numpy().astype(np.float64)[:-1])
    assert torch.equal(z, fresh.encode(xe))
    zr = orc.encode(DIMS, p.cpu().numpy().astype(np.float64)[:-1], x[:1000])
    assert rel(z.cpu().numpy(), zr) < 2e-2


def test_full_size_and_rate():
    """BASELINE configs[1]: 1,000,000 rows in bf16.  Size-independent properties: the gradient of the whole table equals the
    sum of the gradients of its two halves (different workgroup tilings, same math up to fp32 summation order), and the
    kernels hold their rate."""
    n = 1_000_000
    x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
    h, p = handle(orc.formula_params(DIMS, 1))
    g, ga, gb = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    h.fwd_bwd(x, g)
    h.fwd_bwd(x[:n // 2], ga)
    h.fwd_bwd(x[n // 2:], gb)
    assert rel((ga + gb).cpu().numpy(), g.cpu().numpy()) < 1e-4
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        h.fwd_bwd(x, g)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print(f"bf16 fwd_bwd: {ms:.3f} ms per 1M rows = {n / ms / 1e3:.0f} M rows/s")
    assert ms < 1.35, "bf16 training kernels fell off their fast path (profiles: ~1.0 ms per 1M rows)"


@pytest.mark.parametrize("n", [1, 512, 3072, 3073])
def test_small_batches_of_a_bf16_handle_run_the_fp32_kernels(n, data, monkeypatch):
    """Up to 3072 rows a BF16 handle's training calls run the fp32 small-batch kernels (23 us against 34 us per 512-row step):
    gradients at the fp32 bar there, at the bf16 bar above; an optimiser step taken on either path is seen by the other and by the
    bf16 inference kernels (the bf16 fragments are re-rounded on demand)."""
    monkeypatch.delenv("BALER_AMD_BF16_SMALL_ROWS")
    raw, x = data
    flat = orc.formula_params(DIMS, 11)
    h, p = handle(flat)
    g = torch.zeros_like(p)
    xb = torch.as_tensor(x[:n]).cuda()
    h.fwd_bwd(xb, g)
    loss_ref, g_ref = orc.fwd_bwd(DIMS, flat, x[:n])
    gh = g.cpu().numpy().astype(np.float64)
    tol = 1e-5 if n <= 3072 else 2e-2
    assert rel(gh[:-1], g_ref) < tol and abs(gh[-1] - loss_ref) < max(tol, 2e-3 if n > 3072 else 0) * loss_ref
    # train_step on this path, then a large (bf16) and a small (fp32) gradient and an encode with the NEW parameters
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    h.train_step(xb, p, m, v, 1, 1e-2)
    new = p.cpu().numpy().astype(np.float64)[:-1]
    assert rel(new, flat) > 1e-3
    big = torch.as_tensor(x[:8000]).cuda()
    h.fwd_bwd(big, g)
    _, gb = orc.fwd_bwd(DIMS, new, x[:8000])
    assert rel(g.cpu().numpy().astype(np.float64)[:-1], gb) < 2e-2
    small = torch.as_tensor(x[:300]).cuda()
    h.fwd_bwd(small, g)
    _, gs = orc.fwd_bwd(DIMS, new, x[:300])
    assert rel(g.cpu().numpy().astype(np.float64)[:-1], gs) < 1e-5
    assert rel(h.encode(big).cpu().numpy(), orc.encode(DIMS, new, x[:8000])) < 6e-3


@pytest.mark.parametrize("z", [12, 10, 5, 2])
def test_other_latents(z, data):
    """The bf16 kernels (inference and training) at other latent sizes of the compression-ratio knob (latent = ceil(24 / ratio))."""
    raw, x = data
    dims = orc.ae_dims(24, z)
    flat = orc.formula_params(dims, 40 + z)
    h, p = handle(flat, dims=dims)
    assert h.path == "bf16"
    n = 4113
    xd = torch.as_tensor(x[:n]).cuda()
    zr = orc.encode(dims, flat, x[:n])
    assert rel(h.encode(xd).cpu().numpy(), zr) < 2e-2
    assert rel(h.decode(torch.as_tensor(zr).cuda()).cpu().numpy(), orc.decode(dims, flat, zr)) < 2e-2
    g = torch.zeros_like(p)
    h.fwd_bwd(xd, g)
    loss_ref, g_ref = orc.fwd_bwd(dims, flat, x[:n])
    gh = g.cpu().numpy().astype(np.float64)
    assert abs(gh[-1] - loss_ref) < 2e-3 * loss_ref and rel(gh[:-1], g_ref) < 2e-2


def test_wide_model_training_steps_track_the_fp32_run():
    """CFD_dense_AE(2500, 25) on a BF16 handle (the five wide products of the training pass on the bf16 MFMA, csrc/fused.hip
    wide_bf16_train_*_kernel, csrc/generic.hip dw_wide_bf16_k): 20 Adam steps at 600 frames stay within 2 % of the fp32 run's losses,
    one call == two calls bit for bit, the pass is bitwise reproducible, and the handle's encode sees the trained weights."""
    dims = orc.ae_dims(2500, 25)
    flat = orc.formula_params(dims, 5)
    x = torch.as_tensor(synth.cfd_field(2400).reshape(2400, 2500).astype(np.float32)).cuda()
    runs = {}
    for mode in ("fp32", "bf16", "bf16-one-call"):
        h, p = handle(flat, "fp32" if mode == "fp32" else "bf16", dims)
        m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        losses = []
        for t in range(1, 21):
            xb = x[(t - 1) % 4 * 600:][:600]
            if mode == "bf16-one-call":
                h.train_step(xb, p, m, v, t, 1e-3, grads=g)
            else:
                h.fwd_bwd(xb, g)
                h.adam_step(p, g, m, v, t, 1e-3)
            losses.append(float(g[-1]))
        runs[mode] = (np.array(losses), p.clone(), h)
    assert np.max(np.abs(runs["bf16"][0] / runs["fp32"][0] - 1)) < 2e-2
    assert runs["bf16"][0][-1] < 0.9 * runs["bf16"][0][0]
    assert torch.equal(runs["bf16"][1], runs["bf16-one-call"][1])
    h, p = runs["bf16"][2], runs["bf16"][1]
    g1, g2 = torch.zeros_like(p), torch.zeros_like(p)
    h.fwd_bwd(x[:1037], g1)
    h.fwd_bwd(x[:1037], g2)
    assert torch.equal(g1, g2)
    z = h.encode(x[:300], out_dtype=torch.float32)
    zr = orc.encode(dims, p.cpu().numpy().astype(np.float64)[:-1], x[:300].cpu().numpy().astype(np.float64))
    assert rel(z.cpu().numpy(), zr) < 2e-2


def test_wide_model_full_size_and_rate():
    """32,768 CFD frames: the bf16 pass agrees with the fp32 pass on the same handle parameters (loss 1e-3, gradient 5e-3) and is at
    least 1.5x as fast (measured 1.9-2.0x: 0.91 vs 1.80 ms)."""
    dims = orc.ae_dims(2500, 25)
    flat = orc.formula_params(dims, 1)
    n = 32768
    x = torch.rand((n, 2500), dtype=torch.float32, device="cuda")
    res = {}
    for mode in ("fp32", "bf16"):
        h, p = handle(flat, mode, dims)
        g = torch.zeros_like(p)
        h.fwd_bwd(x, g)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(3):
            h.fwd_bwd(x, g)
        e1.record()
        torch.cuda.synchronize()
        res[mode] = (g.cpu().numpy().astype(np.float64), e0.elapsed_time(e1) / 3)
        h.close()
    (gf, tf), (gb, tb) = res["fp32"], res["bf16"]
    print(f"CFD_dense_AE(2500,25) fwd_bwd at {n} frames: fp32 {tf:.3f} ms, bf16 {tb:.3f} ms")
    # dL/drecon stored as bfloat16 (default) or float32 between the launches: both readers round it to bfloat16, so the SAME gradient
    import os
    h, p = handle(flat, "bf16", dims)
    g1, g2 = torch.zeros_like(p), torch.zeros_like(p)
    h.fwd_bwd(x[:4100], g1)
    os.environ["BALER_AMD_BF16_DZ16"] = "0"
    try:
        h.fwd_bwd(x[:4100], g2)
    finally:
        del os.environ["BALER_AMD_BF16_DZ16"]
    assert torch.equal(g1, g2)
    h.close()
    assert abs(gb[-1] - gf[-1]) < 1e-3 * gf[-1] and rel(gb[:-1], gf[:-1]) < 5e-3
    assert np.isfinite(gb).all()
    assert tb < tf / 1.5


@pytest.mark.parametrize("shape,n", [((512, 6), 300001), ((625, 7), 270000)])
def test_wide_model_large_batches(shape, n):
    """More 128-row groups than workgroups (the persistent loops of the bf16 launches take several passes; the shared fragment stage is
    reused across them) and, for 625 columns, the weight gradients of rows that are not 16-byte aligned (dw_short_bf16_k): bf16 pass
    against the fp32 pass of an F32 handle with the same parameters, and bitwise reproducibility."""
    dims = orc.ae_dims(*shape)
    flat = orc.formula_params(dims, 3)
    x = torch.rand((n, shape[0]), dtype=torch.float32, device="cuda")
    res = {}
    for mode in ("fp32", "bf16"):
        h, p = handle(flat, mode, dims)
        g, g2 = torch.zeros_like(p), torch.zeros_like(p)
        h.fwd_bwd(x, g)
        h.fwd_bwd(x, g2)
        assert torch.equal(g, g2)
        res[mode] = g.cpu().numpy().astype(np.float64)
        h.close()
    a, b = res["bf16"], res["fp32"]
    assert np.isfinite(a).all()
    assert abs(a[-1] - b[-1]) < 1e-4 * b[-1] and rel(a[:-1], b[:-1]) < 5e-3
