"""Throughput floors on MI355X: SLACK (30 %) above the steady-state times of the committed bench lines (profiles/r*_bench.json, the
numbers quoted next to every assertion), so a change that drops a kernel off its fast path -- spills, a lost fragment ring, a
fallback to the layer-wise kernels (2-7x) -- fails a test instead of only showing up in the next bench line.  The margin is wider
than the 10-13 % clock-ramp / lease-to-lease variation DESIGN.md section 5 documents: another box, power cap or a busy node must
not fail these without a code regression; the tight numbers are PRINTED for the reader.
Timing as bench.py does it: ~30 ms of the same call first, then the MEDIAN of five event-timed samples on the launch stream.
Synthetic uniform rows."""
import numpy as np
import pytest
import torch

from baler_amd import native
from oracle import c_oracle as orc

pytestmark = pytest.mark.gpu

SLACK = 1.30


def lim(baseline):
    """Upper bound of a time whose committed steady-state value is `baseline`."""
    return baseline * SLACK


def _handle(mode="fp32"):
    dims = orc.ae_dims(24, 15)
    h = native.Handle(dims, mode)
    p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    return h, p


def _ms(fn, reps, warm_ms=30.0, samples=5):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    fn()
    e1.record()
    torch.cuda.synchronize()
    for _ in range(min(200, int(warm_ms / max(e0.elapsed_time(e1), 1e-3)))):
        fn()
    got = []
    for _ in range(samples):
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        got.append(e0.elapsed_time(e1) / reps)
    return sorted(got)[len(got) // 2]


def test_throughput_floors():
    n = 1_000_000
    x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
    h, p = _handle()
    grads = torch.zeros_like(p)
    t_train = _ms(lambda: h.fwd_bwd(x, grads), 5)
    z = h.encode(x)
    t_enc = _ms(lambda: h.encode(x), 5)
    t_dec = _ms(lambda: h.decode(z), 5)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    state = {"t": 0}

    def steps512():
        for i in range(200):
            state["t"] += 1
            h.train_step(x[i * 512:(i + 1) * 512], p, m, v, state["t"], 1e-3)
    t_512 = _ms(steps512, 1, samples=3) / 200
    hb, _ = _handle("bf16")
    t_benc = _ms(lambda: hb.encode(x), 5)
    print(f"fwd_bwd {t_train:.3f} ms, encode {t_enc:.3f} ms, decode {t_dec:.3f} ms, bs512 step {1e3 * t_512:.1f} us, "
          f"bf16 encode {t_benc:.3f} ms per 1M rows")
    assert t_train < lim(3.35), "training pair (bench: 3.35 ms per 1M rows)"
    assert t_enc < lim(0.535) and t_dec < lim(0.513), "fp32 encode / decode (bench: 0.535 / 0.513 ms per 1M rows)"
    assert 1e3 * t_512 < lim(18.2), "small-batch step (bench: 18.2 us)"
    assert t_benc < lim(0.098), "bf16 encode (bench: 0.098 ms per 1M rows)"


def test_round2_kernel_floors():
    """bf16 training pair, fp64 fused small-batch step, fused wide-layer encode / decode (profiles/README.md, round 2)."""
    n = 1_000_000
    x = torch.rand((n, 24), dtype=torch.float64, device="cuda")
    hb, pb = _handle("bf16")
    gb = torch.zeros_like(pb)
    t_b = _ms(lambda: hb.fwd_bwd(x, gb), 5)
    dims = orc.ae_dims(24, 15)
    h64 = native.Handle(dims, "fp64")
    p64 = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
    h64.load_params(p64)
    m, v = torch.zeros_like(p64), torch.zeros_like(p64)
    st = {"t": 0}

    def steps64():
        for i in range(100):
            st["t"] += 1
            h64.train_step(x[i * 512:(i + 1) * 512], p64, m, v, st["t"], 1e-3)
    t_64 = _ms(steps64, 1, samples=3) / 100
    wd = orc.ae_dims(2500, 25)
    hw = native.Handle(wd, "fp32")
    hw.load_params(torch.from_numpy(np.concatenate([orc.formula_params(wd, 1), [0.0]]).astype(np.float32)).cuda())
    xw = torch.rand((32768, 2500), dtype=torch.float32, device="cuda")
    zw = hw.encode(xw)
    t_we = _ms(lambda: hw.encode(xw), 3)
    t_wd = _ms(lambda: hw.decode(zw), 3)
    gw = torch.zeros(orc.nparams(wd) + 1, dtype=torch.float32, device="cuda")
    t_wt = _ms(lambda: hw.fwd_bwd(xw, gw), 3)
    print(f"bf16 fwd_bwd {t_b:.3f} ms per 1M rows, fp64 bs512 step {1e3 * t_64:.1f} us, CFD_dense_AE(2500,25) encode {t_we:.3f} / decode {t_wd:.3f} ms "
          f"/ fwd_bwd {t_wt:.3f} ms per 32768 frames")
    assert t_b < lim(0.745), "bf16 training kernels (r4 bench: 0.745 ms per 1M rows)"
    assert 1e3 * t_64 < lim(41.2), "fp64 fused small-batch step (bench: 41.2 us; layer-wise: 768 us)"
    assert t_we < lim(0.284) and t_wd < lim(0.272), "wide-layer encode / decode (bench: 0.284 / 0.272 ms per 32768 frames; layer-wise 0.66)"
    assert t_wt < lim(1.56), "wide-model training pass (bench: 1.56 ms per 32768 frames; all layer-wise 3.3)"
    hb = native.Handle(wd, "bf16")
    hb.load_params(torch.from_numpy(np.concatenate([orc.formula_params(wd, 1), [0.0]]).astype(np.float32)).cuda())
    xb = torch.rand((131072, 2500), dtype=torch.float32, device="cuda")
    zb = hb.encode(xb, out_dtype=torch.float32)
    t_be = _ms(lambda: hb.encode(xb, out_dtype=torch.float32), 3)
    t_bd = _ms(lambda: hb.decode(zb), 3)
    print(f"bf16 mode, 131072 frames: encode {t_be:.3f} ms, decode {t_bd:.3f} ms")
    assert t_be < lim(0.319) and t_bd < lim(0.439), "bf16 wide-layer encode / decode (round 4: 0.319 / 0.439 ms per 131072 frames = 0.52 / 0.38 of HBM; fp32 1.13 / 1.09)"


def test_round3_kernel_floors():
    """CFD_dense_AE(625, 7) (exafel1 / exafel2 blocks) on the fused wide-layer kernels: encode >= 55 % of the fp32 MFMA peak at
    131072 blocks (300,700 FLOP per block; the layer-wise path it used to fall to reaches 33-47 %)."""
    wd = orc.ae_dims(625, 7)
    hw = native.Handle(wd, "fp32")
    assert hw.path == "fused"
    hw.load_params(torch.from_numpy(np.concatenate([orc.formula_params(wd, 1), [0.0]]).astype(np.float32)).cuda())
    n = 131072
    xw = torch.rand((n, 625), dtype=torch.float32, device="cuda")
    zw = hw.encode(xw)
    t_e = _ms(lambda: hw.encode(xw), 5)
    t_d = _ms(lambda: hw.decode(zw), 5)
    gw = torch.zeros(orc.nparams(wd) + 1, dtype=torch.float32, device="cuda")
    t_t = _ms(lambda: hw.fwd_bwd(xw, gw), 3)
    fe, ft = 300_700 * n / 1e9 / 157.3, 1_554_200 * n / 1e9 / 157.3
    print(f"CFD_dense_AE(625,7), {n} blocks: encode {t_e:.3f} ms = {fe / t_e:.2f} of peak, decode {t_d:.3f} ms = {fe / t_d:.2f}, "
          f"fwd_bwd {t_t:.3f} ms = {ft / t_t:.2f}")
    assert fe / t_e > 0.655 / SLACK, "exafel blocks encode (bench: 0.655 of the fp32 MFMA peak)"
    assert fe / t_d > 0.785 / SLACK and ft / t_t > 0.63 / SLACK, "exafel decode / training pass (bench: 0.785 / 0.63)"


def test_round4_kernel_floors():
    """Round 4: class instantiations for other narrow tables (AE(30, 8): every kernel; AE(48, 12): inference + small-batch training),
    the fp64 fused training step with per-layer tile blocks at 262,144 rows, bf16 encode of the 512-column model."""
    n = 1_000_000
    for (F, Z), path, lim_e, lim_t, lim_s in (((30, 8), "fused", lim(0.55), lim(3.45), lim(24.5)), ((48, 12), "fused", lim(0.75), lim(4.24), lim(26.0))):
        dims = orc.ae_dims(F, Z)
        h = native.Handle(dims, "fp32")
        assert h.path == path
        p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
        h.load_params(p)
        x = torch.rand((n, F), dtype=torch.float64, device="cuda")
        g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        st = {"t": 0}

        def steps():
            for i in range(100):
                st["t"] += 1
                h.train_step(x[i * 512:(i + 1) * 512], p, m, v, st["t"], 1e-3)
        t_e = _ms(lambda: h.encode(x), 5)
        t_s = _ms(steps, 1, samples=3) / 100
        t_t = _ms(lambda: h.fwd_bwd(x, g), 3) if lim_t else float("nan")
        print(f"AE({F},{Z}) [{path}]: encode {t_e:.3f} ms, fwd_bwd {t_t:.3f} ms per 1M rows, bs512 step {1e3 * t_s:.1f} us")
        assert t_e < lim_e, "class encode (round 4: 0.55 / 0.75 ms per 1M rows; layer-wise 2.07)"
        assert 1e3 * t_s < lim_s, "class small-batch step (round 4: 24.5 / 26 us; layer-wise 390)"
        assert lim_t is None or t_t < lim_t, "class throughput pair (round 4: 3.45 ms per 1M rows; layer-wise 11.4)"
        del x
    dims = orc.ae_dims(24, 15)
    h64 = native.Handle(dims, "fp64")
    p64 = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]])).cuda()
    h64.load_params(p64)
    x = torch.rand((262144, 24), dtype=torch.float64, device="cuda")
    g64 = torch.zeros_like(p64)
    t_64 = _ms(lambda: h64.fwd_bwd(x, g64), 3)
    print(f"fp64 fwd_bwd {t_64:.3f} ms per 262144 rows = {357000 * 262144 / t_64 / 1e9 / 78.6:.3f} of the fp64 MFMA peak")
    assert t_64 < lim(2.10), "fp64 fused training step (round 5: 2.10 ms per 262144 rows = 0.57 of the fp64 MFMA peak; round 4: 2.47, round 3: 3.24, layer-wise 5.9)"
    wd = orc.ae_dims(512, 6)
    hb = native.Handle(wd, "bf16")
    hb.load_params(torch.from_numpy(np.concatenate([orc.formula_params(wd, 1), [0.0]]).astype(np.float32)).cuda())
    xb = torch.rand((1 << 20, 512), dtype=torch.float32, device="cuda")
    t_b = _ms(lambda: hb.encode(xb, out_dtype=torch.float32), 3)
    print(f"512-column model, bf16 encode: {t_b:.3f} ms per 1M rows = {2048 * (1 << 20) / t_b / 1e9:.2f} TB/s of rows")
    assert t_b < lim(0.533), "bf16 encode of the 512-column model (round 4: 0.533 ms per 1M rows = 4.0 TB/s of rows; round 3: 2.44 TB/s)"


def test_round5_kernel_floors():
    """Round 5: 64 .. 127-column tables on two states per handle (the wide class for inference and large batches, the small-batch class for
    the 512-row step), the run-time-width wide class on a 900-column model, the 512-row step of the 24-column model after the chain's new
    prologue."""
    n = 1_000_000
    for (F, Z), lim_e, lim_t, lim_s in (((80, 16), lim(0.86), lim(7.0), lim(29.7)), ((127, 31), lim(1.04), lim(7.25), lim(31.0))):
        dims = orc.ae_dims(F, Z)
        h = native.Handle(dims, "fp32")
        assert h.path == "fused"
        p = torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda()
        h.load_params(p)
        x = torch.rand((n, F), dtype=torch.float64, device="cuda")
        g, m, v = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
        st = {"t": 0}

        def steps():
            for i in range(100):
                st["t"] += 1
                h.train_step(x[i * 512:(i + 1) * 512], p, m, v, st["t"], 1e-3)
        t_e = _ms(lambda: h.encode(x), 5)
        t_t = _ms(lambda: h.fwd_bwd(x, g), 3)
        t_s = _ms(steps, 1, samples=3) / 100
        print(f"AE({F},{Z}) [two states]: encode {t_e:.3f} ms, fwd_bwd {t_t:.3f} ms per 1M rows, bs512 step {1e3 * t_s:.1f} us")
        assert t_e < lim_e, "mid-width encode on the wide class (round 5: 0.86 / 1.04 ms per 1M rows; layer-wise 2.1)"
        assert t_t < lim_t, "mid-width fwd_bwd on the wide class (round 5: 7.0 / 7.25 ms per 1M rows; chunked small-batch kernels 9.9 / 10.6, layer-wise 12.6)"
        assert 1e3 * t_s < lim_s, "mid-width 512-row step on the small-batch class (round 5: 29.7 / 31 us; the wide class alone 160, layer-wise 418)"
        del x
    wd = orc.ae_dims(900, 9)
    hw = native.Handle(wd, "fp32")
    assert hw.path == "fused"
    pw = torch.from_numpy(np.concatenate([orc.formula_params(wd, 1), [0.0]]).astype(np.float32)).cuda()
    hw.load_params(pw)
    xw = torch.rand((131072, 900), dtype=torch.float32, device="cuda")
    gw = torch.zeros_like(pw)
    t_we = _ms(lambda: hw.encode(xw, out_dtype=torch.float32), 5)
    t_wt = _ms(lambda: hw.fwd_bwd(xw, gw), 3)
    print(f"AE(900,9) wide class: encode {t_we:.3f} ms, fwd_bwd {t_wt:.3f} ms per 131072 rows")
    assert t_we < lim(0.50) and t_wt < lim(2.60), "wide class on 900 columns (round 5: 0.50 / 2.60 ms per 131072 rows; layer-wise 0.97 / 4.3)"
    h, p = _handle()
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    x = torch.rand((51200, 24), dtype=torch.float64, device="cuda")
    st = {"t": 0}

    def steps24():
        for i in range(100):
            st["t"] += 1
            h.train_step(x[i * 512:(i + 1) * 512], p, m, v, st["t"], 1e-3)
    t_s = _ms(steps24, 1, samples=5) / 100
    print(f"AE(24,15) 512-row step: {1e3 * t_s:.1f} us")
    assert 1e3 * t_s < lim(17.8), "512-row step (round 5: 17.8 us; round 4: 18.2)"
