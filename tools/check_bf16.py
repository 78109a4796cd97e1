"""bf16 inference mode vs the C oracle: error level and throughput (debug / measurement helper)."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from baler_amd import native, synth
from oracle import c_oracle as orc

dims = orc.ae_dims(24, 15)
flat = np.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "g7_c1_model_f32.npz"))["final_params_f32"].astype(np.float64)
raw = synth.cms_rows(10000)
xn = orc.normalize(raw)
feats = orc.find_minmax(raw)
def rel(a, b): return float(np.linalg.norm(np.asarray(a, np.float64) - b) / np.linalg.norm(b))
for mode in ("fp32", "bf16"):
    h = native.Handle(dims, mode)
    p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    xd = torch.from_numpy(xn).cuda()
    z = h.encode(xd)
    zo = orc.encode(dims, flat, xn)
    d = h.decode(torch.from_numpy(zo).cuda())
    do = orc.decode(dims, flat, zo)
    r, loss = h.forward_loss(xd)
    ro = orc.decode(dims, flat, zo)
    lo = float(((ro - xn) ** 2).sum() / 24)
    zr = h.encode(torch.from_numpy(raw).cuda(), features=torch.from_numpy(feats).cuda())
    print(mode, "encode rel", rel(z.cpu().numpy(), zo), "encode(raw+feats) rel", rel(zr.cpu().numpy(), zo), "decode rel", rel(d.cpu().numpy(), do),
          "forward rel", rel(r.cpu().numpy(), ro), "loss rel", abs(loss.item() - lo) / lo, "max abs recon err", np.abs(r.cpu().numpy() - ro).max())
    # ragged / small n
    for n in (1, 17, 63, 64, 65, 1000):
        zz = h.encode(xd[:n]); assert rel(zz.cpu().numpy(), zo[:n]) < 2e-2, n
    # throughput
    big = torch.from_numpy(orc.normalize(synth.cms_rows(1_000_000))).cuda()
    zb = h.encode(big)
    for name, fn in (("encode", lambda: h.encode(big)), ("decode", lambda: h.decode(zb)),
                     ("encode f32 io", None), ("decode f32 io", None)):
        if fn is None:
            b32 = big.float(); z32 = zb.float()
            fn = (lambda: h.encode(b32)) if name.startswith("encode") else (lambda: h.decode(z32))
        fn(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10): fn()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 10
        print(f"   {mode} {name}: {1e6 / dt / 1e9:.2f} G rows/s ({dt * 1e3:.3f} ms per 1M rows)")
