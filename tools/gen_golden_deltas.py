#!/usr/bin/env python3
"""Generate tests/golden/g14_deltas.npz by IMPORTING the reference (authoring container only).

Run:  python tools/gen_golden_deltas.py       (needs /root/reference; writes tests/golden/)

Pins oracle/deltas.py (the error-bounded-deltas side channel) against the
reference's own ``helper.compress(save_error_bounded_deltas=True)`` and ``helper.decompress``:
same flagged (row, col) sets, same float16 deltas, same corrected decoder output.  Only inputs and
expected outputs are stored; the model is the C1 run's final weights (g7_c1_model_f32.npz).

The reference SAVES the side channel with ``np.save`` of ragged Python lists (baler.py:316-338), which
its pinned numpy 1.23 turns into object arrays and numpy >= 1.24 refuses; the generator therefore
calls ``helper.compress`` / ``helper.decompress`` directly and writes the two .gz files itself, as the
object arrays numpy 1.23 would have produced.
"""
import gzip
import os
import sys
import tempfile
import types

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

import numpy as np
import torch

SCRATCH = tempfile.mkdtemp(prefix="baler_golden_deltas_")
os.chdir(SCRATCH)
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from baler.modules import helper as ref_helper  # noqa: E402
from baler.modules import models as ref_models  # noqa: E402

from baler_amd import synth  # noqa: E402
from oracle import c_oracle as orc  # noqa: E402
from oracle import deltas as odeltas  # noqa: E402
from oracle import torch_ref  # noqa: E402

N_ROWS, ROW0, BS, BOUND = 300, 5000, 128, 10


def main():
    dims = orc.ae_dims(24, 15)
    flat = np.load(os.path.join(OUT, "g7_c1_model_f32.npz"))["final_params_f32"].astype(np.float64)
    model = torch_ref.load_flat(ref_models.AE(24, 15), flat)
    model_path = os.path.join(SCRATCH, "model.pt")
    torch.save(model.state_dict(), model_path)
    raw = synth.cms_rows(N_ROWS, row0=ROW0)
    input_path = os.path.join(SCRATCH, "input.npz")
    np.savez(input_path, data=raw, names=synth.CMS_NAMES)
    config = types.SimpleNamespace(
        input_path=input_path, apply_normalization=True, custom_norm=False, data_dimension=1,
        compression_ratio=1.6, batch_size=BS, model_name="AE", model_type="dense",
        save_error_bounded_deltas=True, error_bounded_requirement=BOUND)

    comp, eb_batch, eb_deltas, eb_index = ref_helper.compress(model_path, config)
    assert eb_batch == [0, 1, 2]

    # ---- the restatement must reproduce the reference exactly
    data_n = orc.normalize(raw)
    o_comp, o_batch, o_deltas, o_index = odeltas.compress_with_deltas(dims, flat, data_n, BS, BOUND)
    assert o_batch == eb_batch
    assert np.abs(o_comp - comp).max() < 1e-13
    for k in range(3):
        assert np.array_equal(o_index[k][0], eb_index[k][0]) and np.array_equal(o_index[k][1], eb_index[k][1]), k
        a = np.array(o_deltas[k], dtype=np.float16)
        b = np.array(eb_deltas[k], dtype=np.float16)
        assert a.tobytes() == b.tobytes(), k
    counts = [len(d) for d in eb_deltas]
    print("flagged per batch:", counts, "of", [min(BS, N_ROWS - s) * 24 for s in range(0, N_ROWS, BS)])

    # ---- reference decompress on the side-channel files (object arrays, as numpy 1.23 wrote them)
    comp_path = os.path.join(SCRATCH, "compressed.npz")
    np.savez(comp_path, data=comp, names=synth.CMS_NAMES, normalization_features=np.zeros((2, 24)))
    d_arr = np.empty(3, dtype=object)
    i_arr = np.empty((2, 3), dtype=object)
    for k in range(3):
        d_arr[k] = list(eb_deltas[k])
        i_arr[0, k] = eb_batch[k]
        i_arr[1, k] = eb_index[k]
    deltas_path = os.path.join(SCRATCH, "compressed_deltas.npz.gz")
    index_path = os.path.join(SCRATCH, "compressed_batch_index_metadata.npz.gz")
    with gzip.GzipFile(deltas_path, "w") as f:
        np.save(file=f, arr=d_arr)
    with gzip.GzipFile(index_path, "w") as f:
        np.save(file=f, arr=i_arr)
    dec, _, _ = ref_helper.decompress(model_path, comp_path, deltas_path, index_path, "AE", config, SCRATCH,
                                      raw.shape)
    # restatement of the corrected output
    o_dec = orc.decode(dims, flat, comp)
    plain = o_dec.copy()
    for k, s in enumerate(range(0, N_ROWS, BS)):
        odeltas.apply_deltas(o_dec[s:s + BS], o_deltas[k], o_index[k])
    assert np.abs(o_dec - dec).max() < 1e-13, np.abs(o_dec - dec).max()

    rows = np.concatenate([eb_index[k][0] + k * BS for k in range(3)]).astype(np.uint16)   # global row numbers
    cols = np.concatenate([eb_index[k][1] for k in range(3)]).astype(np.uint8)
    dl = np.concatenate([np.array(eb_deltas[k], dtype=np.float16) for k in range(3)])
    # the corrected elements are now within float16 rounding of the input
    print("max |corrected - x| at flagged:", np.abs(dec[rows, cols] - data_n[rows, cols]).max(),
          " max |plain - x|:", np.abs(plain[rows, cols] - data_n[rows, cols]).max())
    path = os.path.join(OUT, "g14_deltas.npz")
    np.savez(path, n_rows=N_ROWS, row0=ROW0, batch_size=BS, bound=BOUND, counts=np.array(counts),
             rows=rows, cols=cols, deltas=dl, compressed_colsum=comp.sum(axis=0),
             corrected_at_flagged=dec[rows, cols], corrected_colsum=dec.sum(axis=0), corrected_head=dec[:32])
    print(f"wrote g14_deltas.npz: {os.path.getsize(path) / 1024:.1f} KB")


if __name__ == "__main__":
    main()
