"""ctypes wrapper over oracle/baler_oracle.c -- TEST INFRASTRUCTURE ONLY (see oracle/__init__.py)."""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libbaler_oracle.so")
_lib = None

_dp = ctypes.POINTER(ctypes.c_double)
_ip = ctypes.POINTER(ctypes.c_int)


def build(force=False):
    """Compile the C oracle with gcc (no GPU, no reference needed)."""
    src = os.path.join(_HERE, "baler_oracle.c")
    if force or not os.path.exists(_SO) or os.path.getmtime(_SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libbaler_oracle.so"])
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        L = ctypes.CDLL(_SO)
        L.orc_nparams.restype = ctypes.c_long
        L.orc_loss.restype = ctypes.c_double
        L.orc_fwd_bwd.restype = ctypes.c_double
        L.orc_fit_epoch.restype = ctypes.c_double
        L.orc_validate_epoch.restype = ctypes.c_double
        L.orc_emd_rows.restype = ctypes.c_double
        _lib = L
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _dims(dims):
    a = np.ascontiguousarray(dims, dtype=np.int32)
    return a, a.ctypes.data_as(_ip), len(a) - 1


def ae_dims(n_features, z_dim):
    """Layer widths of models.AE / models.CFD_dense_AE (models.py:128-136, 198-206)."""
    return [n_features, 200, 100, 50, z_dim, 50, 100, 200, n_features]


def nparams(dims):
    _, dp, L = _dims(dims)
    return int(lib().orc_nparams(dp, L))


def formula_params(dims, seed=0):
    """Documented init used by every fixture: per tensor in state-dict order,
    default_rng(seed).uniform(-1/sqrt(fan_in), 1/sqrt(fan_in)) (same support as torch's default
    nn.Linear init, models.py:128-136).  Returns the flat fp64 parameter vector."""
    rng = np.random.default_rng(seed)
    parts = []
    for l in range(len(dims) - 1):
        fan_in, out = dims[l], dims[l + 1]
        k = 1.0 / np.sqrt(fan_in)
        parts.append(rng.uniform(-k, k, size=(out, fan_in)).ravel())
        parts.append(rng.uniform(-k, k, size=(out,)))
    return np.concatenate(parts).astype(np.float64)


def find_minmax(data):
    a, ap = _d(data)
    n, c = a.shape
    out = np.empty((2, c), dtype=np.float64)
    lib().orc_find_minmax(ap, ctypes.c_long(n), ctypes.c_int(c), out.ctypes.data_as(_dp))
    return out


def normalize(data):
    a, ap = _d(data)
    n, c = a.shape
    out = np.empty_like(a)
    lib().orc_normalize(ap, ctypes.c_long(n), ctypes.c_int(c), out.ctypes.data_as(_dp))
    return out


def renormalize(norm, minv, rng):
    a, ap = _d(norm)
    n, c = a.shape
    mn, mnp = _d(minv)
    rg, rgp = _d(rng)
    out = np.empty_like(a)
    lib().orc_renormalize(ap, ctypes.c_long(n), ctypes.c_int(c), mnp, rgp, out.ctypes.data_as(_dp))
    return out


def cast_int_cols(data, int_mask):
    a = np.array(data, dtype=np.float64, order="C", copy=True)
    n, c = a.shape
    m = np.ascontiguousarray(int_mask, dtype=np.uint8)
    lib().orc_cast_int_cols(a.ctypes.data_as(_dp), ctypes.c_long(n), ctypes.c_int(c),
                            m.ctypes.data_as(ctypes.POINTER(ctypes.c_ubyte)))
    return a


def _run(fn, dims, params, x, out_dim):
    da, dp, L = _dims(dims)
    p, pp = _d(params)
    a, ap = _d(x)
    n = a.shape[0]
    out = np.empty((n, out_dim), dtype=np.float64)
    fn(dp, ctypes.c_int(L), pp, ap, ctypes.c_long(n), out.ctypes.data_as(_dp))
    return out


def encode(dims, params, x):
    return _run(lib().orc_encode, dims, params, x, dims[(len(dims) - 1) // 2])


def decode(dims, params, z):
    return _run(lib().orc_decode, dims, params, z, dims[-1])


def forward(dims, params, x):
    return _run(lib().orc_forward, dims, params, x, dims[-1])


def loss(x, recon):
    a, ap = _d(x)
    r, rp = _d(recon)
    n, c = a.shape
    return float(lib().orc_loss(ap, rp, ctypes.c_long(n), ctypes.c_int(c)))


def fwd_bwd(dims, params, x):
    """Returns (loss, flat grads) for one batch."""
    da, dp, L = _dims(dims)
    p, pp = _d(params)
    a, ap = _d(x)
    g = np.empty_like(p)
    l = lib().orc_fwd_bwd(dp, ctypes.c_int(L), pp, ap, ctypes.c_long(a.shape[0]),
                          g.ctypes.data_as(_dp))
    return float(l), g


def adam_step(p, g, m, v, t, lr, b1=0.9, b2=0.999, eps=1e-8):
    """In-place Adam step on fp64 numpy arrays (t = step number after increment)."""
    for a in (p, m, v):
        assert a.dtype == np.float64 and a.flags.c_contiguous
    gg, gp = _d(g)
    lib().orc_adam_step(p.ctypes.data_as(_dp), gp, m.ctypes.data_as(_dp), v.ctypes.data_as(_dp),
                        ctypes.c_long(p.size), ctypes.c_long(t), ctypes.c_double(lr),
                        ctypes.c_double(b1), ctypes.c_double(b2), ctypes.c_double(eps))


class FitState:
    """Parameters + Adam state carried across epochs."""

    def __init__(self, dims, params):
        self.dims = list(dims)
        self.params = np.array(params, dtype=np.float64, copy=True)
        self.m = np.zeros_like(self.params)
        self.v = np.zeros_like(self.params)
        self.t = ctypes.c_long(0)


def fit_epoch(state, data, bs, lr):
    """One training.fit epoch; returns (epoch_loss, last_batch_loss)."""
    da, dp, L = _dims(state.dims)
    a, ap = _d(data)
    last = ctypes.c_double(0.0)
    el = lib().orc_fit_epoch(dp, ctypes.c_int(L), state.params.ctypes.data_as(_dp),
                             state.m.ctypes.data_as(_dp), state.v.ctypes.data_as(_dp),
                             ctypes.byref(state.t), ap, ctypes.c_long(a.shape[0]),
                             ctypes.c_long(bs), ctypes.c_double(lr), ctypes.byref(last))
    return float(el), float(last.value)


def validate_epoch(dims, params, data, bs):
    da, dp, L = _dims(dims)
    p, pp = _d(params)
    a, ap = _d(data)
    return float(lib().orc_validate_epoch(dp, ctypes.c_int(L), pp, ap, ctypes.c_long(a.shape[0]),
                                          ctypes.c_long(bs)))


def emd_rows(x, recon):
    a, ap = _d(x)
    r, rp = _d(recon)
    n, c = a.shape
    return float(lib().orc_emd_rows(ap, rp, ctypes.c_long(n), ctypes.c_int(c)))


def activation_means(dims, params, x, max_nodes=200):
    da, dp, L = _dims(dims)
    p, pp = _d(params)
    a, ap = _d(x)
    nact = L - 2
    out = np.empty((nact, max_nodes), dtype=np.float64)
    lib().orc_activation_means(dp, ctypes.c_int(L), pp, ap, ctypes.c_long(a.shape[0]),
                               out.ctypes.data_as(_dp), ctypes.c_int(max_nodes))
    return out


def state_dict_to_flat(sd):
    """Flatten a reference-format state_dict (en1.weight, en1.bias, ... de4.bias) in key order."""
    return np.concatenate([np.asarray(v, dtype=np.float64).ravel() for v in sd.values()])
