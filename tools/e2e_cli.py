#!/usr/bin/env python3
"""End-to-end CLI timing at N rows (npz I/O + PCIe + kernels), the counterpart of SURVEY section 6's
reference numbers (55 k / 27 k / 19 k rows/s for train(1 epoch) / compress / decompress at N = 1 M on 8 CPU cores).
Usage: python tools/e2e_cli.py [N_ROWS] [BATCH_SIZE]"""
import os
import shutil
import subprocess
import sys
import tempfile
import time

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
import numpy as np

from baler_amd import synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
bs = int(sys.argv[2]) if len(sys.argv) > 2 else 512
tmp = tempfile.mkdtemp(prefix="baler_e2e_")
ws = os.path.join(tmp, "workspaces")
shutil.copytree(os.path.join(REPO, "workspaces", "CMS_workspace"), os.path.join(ws, "CMS_workspace"))
open(os.path.join(ws, "__init__.py"), "w").close()
proj = os.path.join(ws, "CMS_workspace", "CMS_project_v1")
for d in ("compressed_output", "decompressed_output", "plotting", "training"):
    os.makedirs(os.path.join(proj, "output", d), exist_ok=True)
os.makedirs(os.path.join(ws, "CMS_workspace", "data"), exist_ok=True)
cfg = os.path.join(proj, "config", "CMS_project_v1_config.py")
src = open(cfg).read().replace("c.epochs = 25", "c.epochs = 1").replace("c.batch_size = 512", f"c.batch_size = {bs}")
open(cfg, "w").write(src)
t0 = time.time()
np.savez(os.path.join(ws, "CMS_workspace", "data", "example_CMS_data.npz"), data=synth.cms_rows(n), names=synth.CMS_NAMES)
print(f"generated {n} rows in {time.time() - t0:.1f}s")
env = dict(os.environ, PYTHONPATH=REPO)
for mode in ("train", "compress", "decompress"):
    t0 = time.time()
    r = subprocess.run([sys.executable, "-m", "baler_amd", "--project", "CMS_workspace", "CMS_project_v1", "--mode", mode],
                       cwd=tmp, env=env, capture_output=True, text=True)
    dt = time.time() - t0
    inner = [l for l in r.stdout.splitlines() if "minutes" in l or "took" in l]
    print(f"{mode:10s} process wall {dt:6.2f}s -> {n / dt / 1e3:8.1f} k rows/s  | rc={r.returncode} | {inner[-1] if inner else r.stderr[-300:]}")
shutil.rmtree(tmp, ignore_errors=True)
