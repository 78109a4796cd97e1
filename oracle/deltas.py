"""TEST INFRASTRUCTURE ONLY -- CPU restatement of Baler's error-bounded-deltas side channel.

Only tests/, tools/gen_golden*.py and bench.py's cpu_baseline leg may import this module; the product
(baler_amd/) never does.

What the reference does (baler/modules/helper.py, reference @ 2024_10_08):

* ``save_error_bounded_requirement`` (helper.py:442-470), called per batch inside ``compress``
  (helper.py:589-606) with the NORMALISED input batch and ``model.decode(model.encode(batch))``:
  ``err = (decoded - data) / data * 100`` in the arrays' dtype; ``+-inf -> 0`` (data == 0); NaN (0/0)
  is left alone because ``== np.nan`` is never true, and a NaN never compares ``> bound``;
  flagged = ``np.where(abs(err) > bound)`` (row-major order); the deltas of the flagged elements are
  ``np.subtract(decoded, data, dtype=np.float16)``: BOTH operands are cast to float16 first and the
  difference is rounded to float16 again.
* ``decompress`` (helper.py:708-718): ``out[row][col] -= delta`` on the decoder output of the same
  batch, before un-normalisation.
* files (baler.py:316-338): ``compressed_deltas.npz.gz`` = gzip(np.save(per-batch lists of float16)),
  ``compressed_batch_index_metadata.npz.gz`` = gzip(np.save(np.array([batches, indices], dtype=object))).

Parity: PINNED -- tools/gen_golden_deltas.py runs the reference's own ``helper.compress`` /
``helper.decompress`` in the authoring container and asserts that this restatement reproduces their
index sets, float16 deltas and corrected output exactly (tests/golden/g14_deltas.npz).

Documented divergence: when NO element of a batch exceeds the bound the reference raises
UnboundLocalError (``deltas`` is only assigned inside the ``if``, helper.py:461-470); this restatement
(and the build) return an empty list for that batch instead.
"""
import numpy as np

from . import c_oracle as orc


def error_bounded_requirement(bound, decoded, data):
    """helper.py:442-470 -> (list of np.float16 deltas, (rows, cols))."""
    decoded = np.asarray(decoded)
    data = np.asarray(data)
    with np.errstate(divide="ignore", invalid="ignore"):
        err = np.divide(np.subtract(decoded, data), data) * 100
    err[(err == np.inf) | (err == -np.inf)] = 0.0
    with np.errstate(invalid="ignore"):
        rows, cols = np.where(abs(err) > bound)
    deltas = []
    if len(rows) > 0:
        diff16 = np.subtract(decoded, data, dtype=np.float16)
        deltas = [diff16[r][c] for r, c in zip(rows, cols)]
    return deltas, (rows, cols)


def compress_with_deltas(dims, params, data_norm, batch_size, bound):
    """The compress loop of helper.py:583-616 with save_error_bounded_deltas=True on normalised rows:
    -> (compressed, batch numbers, per-batch delta lists, per-batch (rows, cols))."""
    comp, batches, all_deltas, all_index = [], [], [], []
    for idx, s in enumerate(range(0, data_norm.shape[0], batch_size)):
        xb = data_norm[s:s + batch_size]
        z = orc.encode(dims, params, xb)
        dec = orc.decode(dims, params, z)
        deltas, index = error_bounded_requirement(bound, dec, xb)
        batches.append(idx)
        all_deltas.append(deltas)
        all_index.append(index)
        comp.append(z)
    return np.concatenate(comp), batches, all_deltas, all_index


def apply_deltas(out_batch, deltas, index):
    """helper.py:708-718 for one batch (in place)."""
    rows, cols = index
    for i in range(len(rows)):
        out_batch[rows[i]][cols[i]] -= deltas[i]
    return out_batch
