#!/usr/bin/env python3
"""Generate tests/golden/g16_edge.npz by IMPORTING the reference (authoring container only).

Run:  python tools/gen_golden_edge.py       (needs /root/reference; writes tests/golden/)

Edge cases of the normalisation the reference does NOT guard (data_processing.py:113-153: no check for
max == min, NaN or +-inf cells): what np.min / np.max / the fp64 division make of them is the behaviour to match.
Two small tables of CMS-like rows:
  table A: ONE +inf cell.  Its column gets max = range = +inf: every finite cell of the column normalises to 0, the
           cell itself to NaN -- one poisoned row, every other row finite (their encodings are compared numerically);
  table B: a constant column (range 0 -> 0/0 = NaN down the column), a constant-zero column, a NaN cell (np.min / np.max
           propagate it: min = range = NaN) and a -inf cell (min = -inf, range = +inf, (x + inf) / inf = NaN).
The script aborts unless the C oracle reproduces the reference bit for bit (NaN positions included).
"""
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

import numpy as np
import torch

os.chdir(tempfile.mkdtemp(prefix="baler_golden_edge_"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from baler.modules import data_processing as ref_dp  # noqa: E402
from baler.modules import helper as ref_helper  # noqa: E402
from baler.modules import models as ref_models  # noqa: E402

from baler_amd import synth  # noqa: E402
from oracle import c_oracle as orc  # noqa: E402
from oracle import torch_ref  # noqa: E402

SEED = 31


def same(a, b):
    """bit-for-bit where finite, same NaN / inf pattern elsewhere"""
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return a.shape == b.shape and np.array_equal(np.isnan(a), np.isnan(b)) and np.array_equal(a[~np.isnan(a)], b[~np.isnan(b)])


def main():
    np.seterr(all="ignore")
    raw_a = synth.cms_rows(41, row0=500)
    raw_a[11, 9] = np.inf
    raw_b = synth.cms_rows(37, row0=900)
    raw_b[:, 3] = 2.5            # constant column: range 0
    raw_b[:, 15] = 0.0           # constant zero
    raw_b[7, 5] = np.nan
    raw_b[2, 12] = -np.inf
    out = {}
    for tag, raw in (("a", raw_a), ("b", raw_b)):
        feats = ref_dp.find_minmax(raw)
        normed = ref_helper.normalize(raw, False)
        renorm = ref_helper.renormalize(normed, feats[0], feats[1])
        assert same(orc.find_minmax(raw), feats), f"oracle find_minmax, table {tag}"
        assert same(orc.normalize(raw), normed), f"oracle normalize, table {tag}"
        assert same(orc.renormalize(normed, feats[0], feats[1]), renorm), f"oracle renormalize, table {tag}"
        out.update({f"raw_{tag}": raw, f"features_{tag}": feats, f"normalized_{tag}": normed, f"renormalized_{tag}": renorm})
        print(f"table {tag}: NaN cells in normalized = {int(np.isnan(normed).sum())} of {normed.size}, "
              f"non-finite features = {int((~np.isfinite(feats)).sum())}")
    # the model on table A: one poisoned row
    dims = orc.ae_dims(24, 15)
    flat = orc.formula_params(dims, SEED)
    model = torch_ref.load_flat(ref_models.AE(24, 15), flat)
    with torch.no_grad():
        z = model.encode(torch.tensor(out["normalized_a"])).numpy()
    bad = np.isnan(z).any(axis=1)
    assert bad.sum() == 1 and bad[11] and np.isnan(z[11]).all()
    z_orc = orc.encode(dims, flat, out["normalized_a"])
    assert np.array_equal(np.isnan(z_orc), np.isnan(z))
    assert np.linalg.norm(z_orc[~bad] - z[~bad]) <= 1e-13 * np.linalg.norm(z[~bad])
    path = os.path.join(OUT, "g16_edge.npz")
    np.savez(path, seed=SEED, z_a=z, **out)
    print(f"wrote g16_edge.npz: {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
