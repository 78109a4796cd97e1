#!/usr/bin/env python3
"""PCIe-inclusive rates of the compress / decompress data path for an N-row file (default 10 M rows x 24 float64 = 1.9 GB):
the stages of helper.compress / helper.decompress timed one by one on the host clock, file on local disk, page cache warm.
  upload    open_npz_array (memory map) -> pinned double-buffered H2D (hostio.upload_rows)
  minmax    column min/max on the resident rows
  encode    bamd_encode per 4M-row block with the download of block k overlapping the encode of block k+1
  whole     everything above, file to host array: the PCIe-inclusive compress rate (bench.py's `value` excludes PCIe)
Usage: python tools/e2e_pcie.py [N_ROWS]"""
import os
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np
import torch

from baler_amd import hostio, native, synth
from baler_amd.modules import models

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10_000_000
tmp = tempfile.mkdtemp(prefix="baler_pcie_")
path = os.path.join(tmp, "data.npz")
t0 = time.time()
chunks = [synth.cms_rows(min(1_000_000, n - r), row0=r) for r in range(0, n, 1_000_000)]
np.savez(path, data=np.concatenate(chunks), names=synth.CMS_NAMES)
del chunks
print(f"file: {n} rows, {os.path.getsize(path) / 1e9:.2f} GB, written in {time.time() - t0:.1f}s")
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = models.AE(24, 15, mode="fp32").to(dev)
h = model.handle()


def sync():
    torch.cuda.synchronize()


for rep in range(2):   # second pass: page cache and pinned staging warm
    sync(); t0 = time.perf_counter()
    src = hostio.open_npz_array(path, "data")
    x = hostio.upload_rows(src, None, dev)
    sync(); t1 = time.perf_counter()
    feats = native.minmax(x)
    sync(); t2 = time.perf_counter()
    out = torch.empty((n, 15), dtype=torch.float64, device=dev)
    ready = []
    B = 1 << 22
    for s in range(0, n, B):
        e = min(s + B, n)
        h.encode(x[s:e], features=feats, out=out[s:e])
        ev = torch.cuda.Event(); ev.record(); ready.append((e, ev))
    z = hostio.download_rows(out, ready=ready)
    sync(); t3 = time.perf_counter()
    # decompress direction: latent rows up, decode (+ un-normalise fused), decoded table down
    zd = hostio.upload_rows(z, None, dev)
    dec = torch.empty((n, 24), dtype=torch.float64, device=dev)
    ready = []
    for s in range(0, n, B):
        e = min(s + B, n)
        h.decode(zd[s:e], features=feats, out=dec[s:e])
        ev = torch.cuda.Event(); ev.record(); ready.append((e, ev))
    back = hostio.download_rows(dec, ready=ready)
    sync(); t4 = time.perf_counter()
    gb = n * 192 / 1e9
    print(f"pass {rep}: upload {t1 - t0:.3f}s ({gb / (t1 - t0):.1f} GB/s)  minmax {1e3 * (t2 - t1):.1f} ms  encode+download {t3 - t2:.3f}s "
          f"({n * 120 / 1e9 / (t3 - t2):.1f} GB/s D2H)  | compress file->host {n / (t3 - t0) / 1e6:.1f} M rows/s PCIe-inclusive  "
          f"| decompress host->host {n / (t4 - t3) / 1e6:.1f} M rows/s PCIe-inclusive")
    del x, out, zd, dec, z, back      # (freeing the host arrays -- ~50 ms per 1.2 GB -- stays outside the next pass's timed regions)
# the reference's way for comparison: pageable .to(device) of the whole table, .cpu().numpy() of the result
sync(); t0 = time.perf_counter()
xt = torch.from_numpy(np.load(path)["data"]).to(dev)
sync(); t1 = time.perf_counter()
zt = h.encode(xt, features=native.minmax(xt)).cpu().numpy()
sync(); t2 = time.perf_counter()
print(f"pageable path: np.load + .to() {t1 - t0:.3f}s, encode + .cpu().numpy() {t2 - t1:.3f}s -> {n / (t2 - t0) / 1e6:.1f} M rows/s")
import shutil
shutil.rmtree(tmp, ignore_errors=True)
