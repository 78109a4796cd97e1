"""Counter-based synthetic inputs.

The CMS example file ``workspaces/CMS_workspace/data/example_CMS_data.npz`` is absent from the
reference checkout (``.MISSING_LARGE_BLOBS``), so every CMS-shaped run uses this generator.  It is
counter based -- value(row, col) depends only on (seed, row, col) -- so any row range can be
produced independently on any rank without materialising the rest.

CMS-like 24 columns following the reference's ``type_list``
(workspaces/CMS_workspace/CMS_project_v1/config/CMS_project_v1_config.py:38-63):
float columns 0-11 and 19-21 are log-normal with mixed scales, "int" columns 12-18 and 22-23 are
small non-negative counts.
"""
import numpy as np

CMS_SEED = 20241008
CMS_NCOLS = 24
CMS_INT_COLS = (12, 13, 14, 15, 16, 17, 18, 22, 23)
CMS_TYPE_LIST = ["int" if c in CMS_INT_COLS else "float64" for c in range(CMS_NCOLS)]
CMS_NAMES = np.array([f"recoPFJets_ak5PFJets_RECO_col{c:02d}" for c in range(CMS_NCOLS)])

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    x = np.asarray(x, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = x + np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, row0, n_rows, n_cols):
    """u(row, col) in (0,1): top 53 bits of splitmix64(seed ^ (row*n_cols + col)), offset by 2^-54."""
    rows = np.arange(row0, row0 + n_rows, dtype=np.uint64)[:, None]
    cols = np.arange(n_cols, dtype=np.uint64)[None, :]
    with np.errstate(over="ignore"):
        ctr = rows * np.uint64(n_cols) + cols
    bits = splitmix64(np.uint64(seed) ^ ctr)
    return ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def _norm_ppf(u):
    """Acklam's rational approximation of the standard normal quantile (|rel err| < 1.2e-9);
    kept dependency-free so the same formula can be evaluated anywhere."""
    a = [-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
         1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00]
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02,
         6.680131188771972e+01, -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00,
         -2.549732539343734e+00, 4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00,
         3.754408661907416e+00]
    u = np.asarray(u, dtype=np.float64)
    out = np.empty_like(u)
    lo = u < 0.02425
    hi = u > 1 - 0.02425
    mid = ~(lo | hi)
    q = np.sqrt(-2 * np.log(u[lo]))
    out[lo] = (((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / \
        ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    q = np.sqrt(-2 * np.log(1 - u[hi]))
    out[hi] = -(((((c[0] * q + c[1]) * q + c[2]) * q + c[3]) * q + c[4]) * q + c[5]) / \
        ((((d[0] * q + d[1]) * q + d[2]) * q + d[3]) * q + 1)
    q = u[mid] - 0.5
    r = q * q
    out[mid] = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / \
        (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
    return out


def cms_rows(n_rows, row0=0, seed=CMS_SEED):
    """(n_rows, 24) float64 CMS-like rows [row0, row0+n_rows)."""
    u = uniform01(seed, row0, n_rows, CMS_NCOLS)
    out = np.empty_like(u)
    for c in range(CMS_NCOLS):
        if c in CMS_INT_COLS:
            out[:, c] = np.floor(-np.log(1.0 - u[:, c]) * (3 + c % 5))
        else:
            out[:, c] = np.exp(_norm_ppf(u[:, c])) * 10.0 ** ((c % 4) - 1)
    return out


def cfd_field(n_frames, h=50, w=50, seed=7):
    """(n, h, w) float64 smooth field in roughly [-0.011, 0.048] like the shipped CFD sample."""
    t = np.arange(n_frames, dtype=np.float64)[:, None, None]
    y = np.arange(h, dtype=np.float64)[None, :, None]
    x = np.arange(w, dtype=np.float64)[None, None, :]
    base = 0.0185 + 0.02 * np.sin(0.13 * x + 0.05 * t + 0.1 * seed) * np.cos(0.11 * y - 0.03 * t)
    ripple = 0.0095 * np.sin(0.37 * x * y / (h + w) + 0.2 * t)
    return base + ripple


def wide_rows(n_rows, n_cols, row0=0, seed=512):
    """(n_rows, n_cols) uniform(0,1) float64 rows for the wide tabular config."""
    return uniform01(seed, row0, n_rows, n_cols)


def cms_rows_torch(n_rows, row0=0, seed=CMS_SEED, device="cuda", chunk=2_000_000):
    """The same counter-based generator evaluated on a torch device (int64 arithmetic wraps like uint64; logical
    shifts by masking), chunk by chunk into one (n_rows, 24) float64 tensor: what bench.py uses for the 12.5 M-row
    shard of BASELINE configs[2], which would take a minute of numpy per rank.  Same formula as cms_rows; the last
    bits of exp / log may differ between libraries, so parity tests keep using cms_rows."""
    import torch

    def lsr(z, k):
        return (z >> k) & ((1 << (64 - k)) - 1)

    def i64(v):
        v &= 0xFFFFFFFFFFFFFFFF
        return v - (1 << 64) if v >= (1 << 63) else v

    out = torch.empty((n_rows, CMS_NCOLS), dtype=torch.float64, device=device)
    cols = torch.arange(CMS_NCOLS, dtype=torch.int64, device=device)[None, :]
    a = torch.tensor([-3.969683028665376e+01, 2.209460984245205e+02, -2.759285104469687e+02,
                      1.383577518672690e+02, -3.066479806614716e+01, 2.506628277459239e+00], dtype=torch.float64)
    b = [-5.447609879822406e+01, 1.615858368580409e+02, -1.556989798598866e+02, 6.680131188771972e+01, -1.328068155288572e+01]
    c = [-7.784894002430293e-03, -3.223964580411365e-01, -2.400758277161838e+00, -2.549732539343734e+00,
         4.374664141464968e+00, 2.938163982698783e+00]
    d = [7.784695709041462e-03, 3.224671290700398e-01, 2.445134137142996e+00, 3.754408661907416e+00]
    a = [float(v) for v in a]
    is_int = torch.tensor([cc in CMS_INT_COLS for cc in range(CMS_NCOLS)], device=device)[None, :]
    int_scale = torch.tensor([3.0 + cc % 5 for cc in range(CMS_NCOLS)], dtype=torch.float64, device=device)[None, :]
    flt_scale = torch.tensor([10.0 ** ((cc % 4) - 1) for cc in range(CMS_NCOLS)], dtype=torch.float64, device=device)[None, :]
    for lo in range(0, n_rows, chunk):
        hi = min(n_rows, lo + chunk)
        rows = torch.arange(row0 + lo, row0 + hi, dtype=torch.int64, device=device)[:, None]
        z = (rows * CMS_NCOLS + cols) ^ i64(seed)
        z = z + i64(0x9E3779B97F4A7C15)
        z = (z ^ lsr(z, 30)) * i64(0xBF58476D1CE4E5B9)
        z = (z ^ lsr(z, 27)) * i64(0x94D049BB133111EB)
        z = z ^ lsr(z, 31)
        u = (lsr(z, 11).to(torch.float64) + 0.5) * (1.0 / 9007199254740992.0)
        ql = torch.sqrt(-2 * torch.log(u.clamp(max=0.5)))
        lo_v = (((((c[0] * ql + c[1]) * ql + c[2]) * ql + c[3]) * ql + c[4]) * ql + c[5]) / ((((d[0] * ql + d[1]) * ql + d[2]) * ql + d[3]) * ql + 1)
        qh = torch.sqrt(-2 * torch.log((1 - u).clamp(max=0.5)))
        hi_v = -(((((c[0] * qh + c[1]) * qh + c[2]) * qh + c[3]) * qh + c[4]) * qh + c[5]) / ((((d[0] * qh + d[1]) * qh + d[2]) * qh + d[3]) * qh + 1)
        q = u - 0.5
        r = q * q
        mid = (((((a[0] * r + a[1]) * r + a[2]) * r + a[3]) * r + a[4]) * r + a[5]) * q / (((((b[0] * r + b[1]) * r + b[2]) * r + b[3]) * r + b[4]) * r + 1)
        ppf = torch.where(u < 0.02425, lo_v, torch.where(u > 1 - 0.02425, hi_v, mid))
        out[lo:hi] = torch.where(is_int, torch.floor(-torch.log(1.0 - u) * int_scale), torch.exp(ppf) * flt_scale)
    return out
