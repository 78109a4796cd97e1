# needs a debug build of the library: make -C baler_amd/csrc clean all HIPFLAGS+=-DBAMD_DEBUG (bamd_debug_copy_imgs)
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
dims = orc.ae_dims(24, 15)
flat = orc.formula_params(dims, 7)
x = orc.normalize(synth.cms_rows(1000))
def imgs(env, n):
    os.environ["BALER_AMD_LAT4_ROWS"] = env
    h = native.Handle(dims, "fp32")
    p = torch.as_tensor(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    g = torch.zeros_like(p)
    h.fwd_bwd(torch.as_tensor(x[:n]).cuda(), g)
    torch.cuda.synchronize()
    nfl = 26624 * ((n + 15) // 16)
    out = torch.zeros(nfl, dtype=torch.float32, device="cuda")
    L = native.lib()
    L.bamd_debug_copy_imgs.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t]
    rc = L.bamd_debug_copy_imgs(h._h, ctypes.c_void_p(out.data_ptr()), nfl * 4)
    return out.cpu().numpy().reshape(-1, 16), rc
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
a, rc = imgs("1024", n)
b, _ = imgs("0", n)
print("rc", rc, a.shape)
# slot offsets: X images 0..7 then dZ 0..7
def tiles(d): return (d + 15) // 16
d = dims
xr = [16 * tiles(d[l] + 1) for l in range(8)]
zr = [16 * tiles(d[l + 1]) for l in range(8)]
nb = (n + 15) // 16
a3, b3 = a.reshape(nb, -1, 16), b.reshape(nb, -1, 16)
off = 0
for name, rows in [(f"X{l}", xr[l]) for l in range(8)] + [(f"dZ{l}", zr[l]) for l in range(8)]:
    A, B = a3[:, off:off + rows], b3[:, off:off + rows]
    bad = np.argwhere(~np.isclose(A, B, rtol=1e-4, atol=1e-9))
    print(f"{name:4s} slots {off:5d}..{off + rows:5d}  max|diff| {np.abs(A - B).max():.3e}  mismatches {len(bad)}" + (f"  first {bad[:4].tolist()} a={A[tuple(bad[0])]:.4g} b={B[tuple(bad[0])]:.4g}" if len(bad) else ""))
    off += rows
np.set_printoptions(precision=4, linewidth=200)
print("x[:4, :6]\n", x[:4, :6])
print("lat4 X0 slots 0..5, cols 0..7\n", a[:6, :8])
print("lat2 X0 slots 0..5, cols 0..7\n", b[:6, :8])
