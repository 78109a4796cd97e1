"""ctypes binding of libbaler_amd.so (include/baler_amd.h).

PyTorch is used only as the owner of device memory and of the current HIP stream; every call
below passes raw device pointers (``tensor.data_ptr()``) across the C ABI.  There is NO CPU or
PyTorch fallback: if the library is missing, fails to load, or no gfx950 device is present, the
calls raise ``NativeError``.
"""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("BALER_AMD_LIB", os.path.join(_HERE, "libbaler_amd.so"))  # override: kernel A/B experiments

F32, F64 = 0, 1
MODE_F32, MODE_F64, MODE_BF16 = 0, 1, 2
MODE_NAMES = {"fp32": MODE_F32, "f32": MODE_F32, "fp64": MODE_F64, "f64": MODE_F64, "bf16": MODE_BF16}

# every symbol include/baler_amd.h declares (tests check that the library exports all of them)
SYMBOLS = (
    "bamd_abi_version", "bamd_last_error", "bamd_device_count", "bamd_create", "bamd_destroy",
    "bamd_param_count", "bamd_mode_of", "bamd_load_params", "bamd_minmax", "bamd_normalize",
    "bamd_renormalize", "bamd_encode", "bamd_decode", "bamd_forward_loss", "bamd_fwd_bwd",
    "bamd_adam_step", "bamd_train_step", "bamd_emd_rows", "bamd_activation_means",
    "bamd_error_deltas", "bamd_apply_deltas", "bamd_fwd_bwd_latent", "bamd_swd", "bamd_col_minmax", "bamd_path_of",
    "bamd_train_epoch", "bamd_comm_unique_id", "bamd_comm_init", "bamd_comm_attach", "bamd_comm_release", "bamd_comm_world",
    "bamd_allreduce_sum", "bamd_train_epoch_dp",
)


class NativeError(RuntimeError):
    pass


class AdamHP(ctypes.Structure):
    _fields_ = [("step", ctypes.c_int64), ("lr", ctypes.c_double), ("beta1", ctypes.c_double),
                ("beta2", ctypes.c_double), ("eps", ctypes.c_double)]


_lib = None


def lib():
    """Load libbaler_amd.so (built in-tree by ``make -C baler_amd/csrc`` / ``__graft_entry__.build``)."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise NativeError(f"{LIB_PATH} is missing: build it with `make -C baler_amd/csrc` "
                          "(there is no CPU fallback for the hot path)")
    try:
        L = ctypes.CDLL(LIB_PATH)
    except OSError as e:  # pragma: no cover
        raise NativeError(f"cannot load {LIB_PATH}: {e}") from e
    vp, i64, ci, dbl = ctypes.c_void_p, ctypes.c_int64, ctypes.c_int, ctypes.c_double
    L.bamd_abi_version.restype = ci
    L.bamd_last_error.restype = ctypes.c_char_p
    L.bamd_device_count.restype = ci
    L.bamd_create.argtypes = [ctypes.POINTER(ci), ci, ci, ci, ctypes.POINTER(vp)]
    L.bamd_destroy.argtypes = [vp]
    L.bamd_destroy.restype = None
    L.bamd_param_count.argtypes = [vp]
    L.bamd_param_count.restype = i64
    L.bamd_mode_of.argtypes = [vp]
    L.bamd_mode_of.restype = ci
    L.bamd_load_params.argtypes = [vp, vp, ci, vp]
    L.bamd_minmax.argtypes = [vp, ci, i64, ci, vp, vp]
    L.bamd_col_minmax.argtypes = [vp, ci, i64, ci, vp, vp]
    L.bamd_path_of.argtypes = [vp]
    L.bamd_normalize.argtypes = [vp, ci, i64, ci, vp, vp, ci, vp]
    L.bamd_renormalize.argtypes = [vp, ci, i64, ci, vp, vp, vp, vp]
    L.bamd_encode.argtypes = [vp, vp, ci, i64, vp, vp, ci, vp]
    L.bamd_decode.argtypes = [vp, vp, ci, i64, vp, vp, vp, ci, vp]
    L.bamd_forward_loss.argtypes = [vp, vp, ci, i64, vp, vp, ci, vp, vp]
    L.bamd_fwd_bwd.argtypes = [vp, vp, ci, i64, vp, vp, vp]
    L.bamd_adam_step.argtypes = [vp, vp, vp, vp, vp, ctypes.POINTER(AdamHP), vp, vp]
    L.bamd_train_step.argtypes = [vp, vp, ci, i64, vp, vp, vp, vp, vp, ctypes.POINTER(AdamHP), vp, vp]
    L.bamd_train_epoch.argtypes = [vp, vp, ci, i64, i64, vp, vp, vp, vp, vp, ctypes.POINTER(AdamHP), vp, ctypes.POINTER(i64), vp]
    L.bamd_comm_unique_id.argtypes = [vp]
    L.bamd_comm_init.argtypes = [vp, vp, ci, ci]
    L.bamd_comm_attach.argtypes = [vp, vp, ci]
    L.bamd_comm_release.argtypes = [vp]
    L.bamd_comm_world.argtypes = [vp]
    L.bamd_allreduce_sum.argtypes = [vp, vp, ci, i64, vp]
    L.bamd_train_epoch_dp.argtypes = [vp, vp, ci, ctypes.POINTER(i64), i64, vp, vp, vp, vp, vp, ctypes.POINTER(AdamHP), vp, vp]
    L.bamd_emd_rows.argtypes = [vp, vp, ci, i64, ci, vp, vp]
    L.bamd_activation_means.argtypes = [vp, vp, ci, i64, vp, vp, ci, vp]
    L.bamd_error_deltas.argtypes = [vp, vp, ci, i64, dbl, vp, vp, vp]
    L.bamd_fwd_bwd_latent.argtypes = [vp, vp, ci, i64, vp, vp, vp, vp]
    L.bamd_swd.argtypes = [vp, vp, vp, ci, i64, ci, ci, dbl, vp, vp, vp]
    L.bamd_apply_deltas.argtypes = [vp, ci, ci, vp, vp, vp, i64, vp]
    for name in SYMBOLS:
        getattr(L, name)
    _lib = L
    return L


def _check(rc, what):
    if rc != 0:
        msg = lib().bamd_last_error().decode(errors="replace")
        raise NativeError(f"{what} failed (status {rc}): {msg}")


def _dt(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.float64:
        return F64
    raise NativeError(f"unsupported tensor dtype {t.dtype}")


def _dev_tensor(t):
    if not isinstance(t, torch.Tensor) or not t.is_cuda:
        raise NativeError("expected a CUDA/HIP tensor (the hot path has no CPU fallback)")
    if not t.is_contiguous():
        raise NativeError("expected a contiguous tensor")
    return t


def _stream(t=None):
    """The current stream OF THE TENSOR'S DEVICE (not of whichever device happens to be current: a stream handle of
    cuda:0 is not valid for buffers and a handle that live on cuda:1)."""
    dev = t.device if t is not None else None
    return ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)


def _same_device(what, first, *others):
    for t in others:
        if t is not None and t.device != first.device:
            raise NativeError(f"{what}: tensors live on different devices ({first.device} vs {t.device})")


def _ptr(t):
    return ctypes.c_void_p(t.data_ptr()) if t is not None else ctypes.c_void_p(0)


def require_gpu():
    n = lib().bamd_device_count()
    if n <= 0 or not torch.cuda.is_available():
        raise NativeError("no MI355X (gfx950) device visible: the baler_amd hot path has no CPU fallback")
    return n


def comm_unique_id():
    """128 bytes identifying a new RCCL communicator (ncclGetUniqueId); rank 0 creates it and sends it to the other ranks."""
    buf = ctypes.create_string_buffer(128)
    _check(lib().bamd_comm_unique_id(buf), "bamd_comm_unique_id")
    return buf.raw


# ---- handle-free kernels --------------------------------------------------------------------------
def minmax(x):
    """data_processing.find_minmax on device: x (n, c) f32/f64 -> (2, c) f64 [min ; max-min]."""
    x = _dev_tensor(x)
    n, c = x.shape
    out = torch.empty((2, c), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().bamd_minmax(_ptr(x), _dt(x), n, c, _ptr(out), _stream(x)), "bamd_minmax")
    return out


def col_minmax(x):
    """Raw column extrema of this rank's rows: x (n, c) f32/f64 -> (2, c) f64 [min ; max] (bamd_col_minmax); the
    data-parallel form of find_minmax all-reduces row 0 with MIN and row 1 with MAX and takes range = max - min."""
    x = _dev_tensor(x)
    n, c = x.shape
    out = torch.empty((2, c), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().bamd_col_minmax(_ptr(x), _dt(x), n, c, _ptr(out), _stream(x)), "bamd_col_minmax")
    return out


def normalize(x, features, out_dtype=torch.float64):
    x = _dev_tensor(x)
    features = _dev_tensor(features)
    n, c = x.shape
    _same_device("normalize", x, features)
    out = torch.empty((n, c), dtype=out_dtype, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().bamd_normalize(_ptr(x), _dt(x), n, c, _ptr(features), _ptr(out), _dt(out), _stream(x)),
               "bamd_normalize")
    return out


def renormalize(x, features, int_mask=None):
    x = _dev_tensor(x)
    n, c = x.shape
    _same_device("renormalize", x, features, int_mask)
    out = torch.empty((n, c), dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().bamd_renormalize(_ptr(x), _dt(x), n, c, _ptr(features), _ptr(int_mask), _ptr(out),
                                      _stream(x)), "bamd_renormalize")
    return out


def emd_rows(x, recon):
    x = _dev_tensor(x)
    recon = _dev_tensor(recon)
    if x.dtype != recon.dtype or x.shape != recon.shape:
        raise NativeError("emd_rows: x and recon must have the same dtype and shape")
    _same_device("emd_rows", x, recon)
    out = torch.empty(1, dtype=torch.float64, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().bamd_emd_rows(_ptr(x), _ptr(recon), _dt(x), x.shape[0], x.shape[1], _ptr(out), _stream(x)),
               "bamd_emd_rows")
    return out


def swd(z, prior, proj, reg_weight):
    """utils.compute_swd forward + backward: -> (loss float64[1], dz like z).  z, prior (n, d); proj (s, d) unit rows."""
    z, prior, proj = _dev_tensor(z), _dev_tensor(prior), _dev_tensor(proj)
    if not (z.dtype == prior.dtype == proj.dtype) or z.shape != prior.shape or proj.shape[1] != z.shape[1]:
        raise NativeError("swd: z/prior (n, d) and proj (s, d) must share dtype and latent size")
    _same_device("swd", z, prior, proj)
    loss = torch.empty(1, dtype=torch.float64, device=z.device)
    dz = torch.empty_like(z)
    with torch.cuda.device(z.device):
        _check(lib().bamd_swd(_ptr(z), _ptr(prior), _ptr(proj), _dt(z), z.shape[0], z.shape[1], proj.shape[0],
                              float(reg_weight), _ptr(loss), _ptr(dz), _stream(z)), "bamd_swd")
    return loss, dz


def error_deltas(x, recon, bound):
    """helper.save_error_bounded_requirement for a whole table: -> (flags uint8 (n, c), deltas float16 (n, c))."""
    x = _dev_tensor(x)
    recon = _dev_tensor(recon)
    if x.dtype != recon.dtype or x.shape != recon.shape:
        raise NativeError("error_deltas: x and recon must have the same dtype and shape")
    _same_device("error_deltas", x, recon)
    flags = torch.empty(x.shape, dtype=torch.uint8, device=x.device)
    deltas = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    with torch.cuda.device(x.device):
        _check(lib().bamd_error_deltas(_ptr(x), _ptr(recon), _dt(x), x.numel(), float(bound), _ptr(flags), _ptr(deltas),
                                       _stream(x)), "bamd_error_deltas")
    return flags, deltas


def apply_deltas(out, rows, cols, deltas):
    """out[rows[i], cols[i]] -= deltas[i] in place (rows int64, cols int32, deltas float16; device tensors)."""
    out = _dev_tensor(out)
    rows, cols, deltas = _dev_tensor(rows), _dev_tensor(cols), _dev_tensor(deltas)
    if rows.dtype != torch.int64 or cols.dtype != torch.int32 or deltas.dtype != torch.float16:
        raise NativeError("apply_deltas: rows int64, cols int32, deltas float16 expected")
    if not (rows.numel() == cols.numel() == deltas.numel()):
        raise NativeError("apply_deltas: rows, cols and deltas must have the same length")
    _same_device("apply_deltas", out, rows, cols, deltas)
    with torch.cuda.device(out.device):
        _check(lib().bamd_apply_deltas(_ptr(out), _dt(out), out.shape[1], _ptr(rows), _ptr(cols), _ptr(deltas),
                                       rows.numel(), _stream(out)), "bamd_apply_deltas")
    return out


# ---- model handle ---------------------------------------------------------------------------------
class Handle:
    """Owns a bamd_handle*; parameters and optimiser state stay in caller-owned torch tensors."""

    def __init__(self, dims, mode="fp32", device=None):
        require_gpu()
        self.dims = [int(d) for d in dims]
        self.mode = MODE_NAMES[mode] if isinstance(mode, str) else int(mode)
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else device)
        arr = (ctypes.c_int * len(self.dims))(*self.dims)
        h = ctypes.c_void_p()
        _check(lib().bamd_create(arr, len(self.dims) - 1, self.mode, self.device.index, ctypes.byref(h)),
               "bamd_create")
        self._h = h
        self.nparams = int(lib().bamd_param_count(h))
        self.param_dtype = torch.float64 if self.mode == MODE_F64 else torch.float32
        # the mode the library computes in: "bf16" asked of a shape without bf16 kernels is served in float32 (notice on stderr)
        self.compute_mode = int(lib().bamd_mode_of(h))

    def close(self):
        if getattr(self, "_h", None):
            lib().bamd_destroy(self._h)
            self._h = None

    def __del__(self):  # pragma: no cover
        try:
            self.close()
        except Exception:
            pass

    def _s(self):
        """Current torch stream of THE HANDLE'S device (the library makes that device current for the call)."""
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _mine(self, *tensors):
        for t in tensors:
            if t is not None and t.device != self.device:
                raise NativeError(f"tensor on {t.device} passed to a handle that lives on {self.device}")

    @property
    def z_dim(self):
        return self.dims[(len(self.dims) - 1) // 2]

    @property
    def path(self):
        """"fused" | "bf16" | "generic" | "fused-infer" (fused encode / decode / validation, layer-wise training): which kernels
        serve this shape's throughput calls (bamd_path_of)."""
        return {0: "generic", 1: "fused", 2: "bf16", 3: "fused-infer"}[int(lib().bamd_path_of(self._h))]

    def load_params(self, flat):
        flat = _dev_tensor(flat)
        self._mine(flat)
        if flat.numel() < self.nparams:
            raise NativeError("parameter vector too short")
        _check(lib().bamd_load_params(self._h, _ptr(flat), _dt(flat), self._s()), "bamd_load_params")

    def _out(self, out, rows, cols, dtype, device):
        """Caller-provided output (a contiguous row block of a preallocated result) or a fresh tensor."""
        if out is None:
            return torch.empty((rows, cols), dtype=dtype, device=device)
        _dev_tensor(out)
        self._mine(out)
        if tuple(out.shape) != (rows, cols):
            raise NativeError(f"out has shape {tuple(out.shape)}, expected {(rows, cols)}")
        return out

    def encode(self, x, features=None, out_dtype=None, out=None):
        x = _dev_tensor(x)
        self._mine(x, features)
        out = self._out(out, x.shape[0], self.z_dim, out_dtype or x.dtype, x.device)
        _check(lib().bamd_encode(self._h, _ptr(x), _dt(x), x.shape[0], _ptr(features), _ptr(out), _dt(out),
                                 self._s()), "bamd_encode")
        return out

    def decode(self, z, features=None, int_mask=None, out_dtype=None, out=None):
        z = _dev_tensor(z)
        self._mine(z, features, int_mask)
        out = self._out(out, z.shape[0], self.dims[-1], out_dtype or z.dtype, z.device)
        _check(lib().bamd_decode(self._h, _ptr(z), _dt(z), z.shape[0], _ptr(features), _ptr(int_mask),
                                 _ptr(out), _dt(out), self._s()), "bamd_decode")
        return out

    def forward_loss(self, x, features=None, want_recon=True, loss_out=None):
        x = _dev_tensor(x)
        self._mine(x, features, loss_out)
        recon = torch.empty_like(x) if want_recon else None
        loss = loss_out if loss_out is not None else torch.empty(1, dtype=torch.float64, device=x.device)
        _check(lib().bamd_forward_loss(self._h, _ptr(x), _dt(x), x.shape[0], _ptr(features), _ptr(recon),
                                       _dt(recon) if recon is not None else F32, _ptr(loss), self._s()),
               "bamd_forward_loss")
        return recon, loss

    def fwd_bwd(self, x, grads, features=None):
        x = _dev_tensor(x)
        grads = _dev_tensor(grads)
        self._mine(x, grads, features)
        if grads.dtype != self.param_dtype or grads.numel() < self.nparams + 1:
            raise NativeError("grads must hold param_count+1 elements of the handle's parameter type")
        _check(lib().bamd_fwd_bwd(self._h, _ptr(x), _dt(x), x.shape[0], _ptr(features), _ptr(grads), self._s()),
               "bamd_fwd_bwd")

    def fwd_bwd_latent(self, x, latent_grad, grads, features=None):
        """fwd_bwd with dL/dz of a latent regulariser (n, z_dim; handle parameter type) injected at the bottleneck."""
        x = _dev_tensor(x)
        grads = _dev_tensor(grads)
        latent_grad = _dev_tensor(latent_grad)
        self._mine(x, grads, latent_grad, features)
        if grads.dtype != self.param_dtype or grads.numel() < self.nparams + 1:
            raise NativeError("grads must hold param_count+1 elements of the handle's parameter type")
        if latent_grad.dtype != self.param_dtype or tuple(latent_grad.shape) != (x.shape[0], self.z_dim):
            raise NativeError("latent_grad must be (n_rows, z_dim) of the handle's parameter type")
        _check(lib().bamd_fwd_bwd_latent(self._h, _ptr(x), _dt(x), x.shape[0], _ptr(features), _ptr(latent_grad),
                                         _ptr(grads), self._s()), "bamd_fwd_bwd_latent")

    def adam_step(self, params, grads, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, loss_accum=None):
        self._mine(params, grads, m, v, loss_accum)
        for t in (params, grads, m, v):
            _dev_tensor(t)
            if t.dtype != self.param_dtype:
                raise NativeError("optimizer tensors must have the handle's parameter type")
        hp = AdamHP(int(step), float(lr), float(beta1), float(beta2), float(eps))
        _check(lib().bamd_adam_step(self._h, _ptr(params), _ptr(grads), _ptr(m), _ptr(v), ctypes.byref(hp),
                                    _ptr(loss_accum), self._s()), "bamd_adam_step")

    def train_step(self, x, params, m, v, step, lr, beta1=0.9, beta2=0.999, eps=1e-8, loss_accum=None, grads=None,
                   features=None):
        """fwd + loss + bwd + Adam of one batch in one call (= fwd_bwd then adam_step; no all-reduce in between)."""
        x = _dev_tensor(x)
        self._mine(x, params, m, v, grads, loss_accum, features)
        for t in (params, m, v):
            _dev_tensor(t)
            if t.dtype != self.param_dtype:
                raise NativeError("optimizer tensors must have the handle's parameter type")
        if grads is not None and (grads.dtype != self.param_dtype or grads.numel() < self.nparams + 1):
            raise NativeError("grads must hold param_count+1 elements of the handle's parameter type")
        hp = AdamHP(int(step), float(lr), float(beta1), float(beta2), float(eps))
        _check(lib().bamd_train_step(self._h, _ptr(x), _dt(x), x.shape[0], _ptr(features), _ptr(params), _ptr(grads),
                                     _ptr(m), _ptr(v), ctypes.byref(hp), _ptr(loss_accum), self._s()), "bamd_train_step")

    def train_epoch(self, x, batch_size, params, m, v, first_step, lr, beta1=0.9, beta2=0.999, eps=1e-8, loss_accum=None,
                    grads=None, features=None):
        """One epoch of sequential batches over the resident rows `x` in ONE native call (bamd_train_epoch: the batch loop runs
        inside the library, every batch as train_step).  `first_step` = Adam's t of the first batch.  Returns the number of steps."""
        x = _dev_tensor(x)
        self._mine(x, params, m, v, grads, loss_accum, features)
        for t in (params, m, v):
            _dev_tensor(t)
            if t.dtype != self.param_dtype:
                raise NativeError("optimizer tensors must have the handle's parameter type")
        if grads is not None and (grads.dtype != self.param_dtype or grads.numel() < self.nparams + 1):
            raise NativeError("grads must hold param_count+1 elements of the handle's parameter type")
        hp = AdamHP(int(first_step), float(lr), float(beta1), float(beta2), float(eps))
        steps = ctypes.c_int64(0)
        _check(lib().bamd_train_epoch(self._h, _ptr(x), _dt(x), x.shape[0], int(batch_size), _ptr(features), _ptr(params),
                                      _ptr(grads), _ptr(m), _ptr(v), ctypes.byref(hp), _ptr(loss_accum), ctypes.byref(steps),
                                      self._s()), "bamd_train_epoch")
        return int(steps.value)

    # ---- data-parallel training inside the library (RCCL resolved at run time by the library) -----------------------------
    def comm_init(self, unique_id, rank, world):
        """Collective: ncclCommInitRank(world, unique_id, rank) on the handle's device; afterwards train_step / train_epoch_dp run
        fwd_bwd -> ncclAllReduce(sum) -> Adam inside the library.  `unique_id`: the 128 bytes of comm_unique_id() of rank 0."""
        if len(unique_id) != 128:
            raise NativeError("unique_id must be the 128 bytes of comm_unique_id()")
        buf = (ctypes.c_char * 128).from_buffer_copy(bytes(unique_id))
        with torch.cuda.device(self.device):
            _check(lib().bamd_comm_init(self._h, buf, int(rank), int(world)), "bamd_comm_init")

    def comm_attach(self, comm, world=0):
        """Use a communicator the CALLER made (an ncclComm_t as an integer address, from the same RCCL the library resolves); the
        handle never destroys it.  world <= 0: ask ncclCommCount."""
        _check(lib().bamd_comm_attach(self._h, ctypes.c_void_p(int(comm)), int(world)), "bamd_comm_attach")

    def comm_release(self):
        """Detach the communicator (ncclCommDestroy only if comm_init made it); the handle trains single-process again."""
        _check(lib().bamd_comm_release(self._h), "bamd_comm_release")

    @property
    def comm_world(self):
        """Ranks of the communicator attached to this handle (0: none, the handle trains single-process)."""
        return int(lib().bamd_comm_world(self._h))

    def allreduce_sum(self, t):
        """In-place SUM all-reduce of a float32 / float64 tensor over the handle's communicator, on the current stream."""
        t = _dev_tensor(t)
        self._mine(t)
        _check(lib().bamd_allreduce_sum(self._h, _ptr(t), _dt(t), t.numel(), self._s()), "bamd_allreduce_sum")
        return t

    def train_epoch_dp(self, x, batch_rows, params, m, v, first_step, lr, beta1=0.9, beta2=0.999, eps=1e-8, loss_accum=None,
                       grads=None, features=None):
        """One epoch of the data-parallel batch loop in ONE native call: `x` = this rank's rows of every global batch back to back,
        batch_rows[i] = its rows of global batch i (bamd_train_epoch_dp).  Returns the number of steps."""
        x = _dev_tensor(x)
        self._mine(x, params, m, v, grads, loss_accum, features)
        for t in (params, m, v):
            _dev_tensor(t)
            if t.dtype != self.param_dtype:
                raise NativeError("optimizer tensors must have the handle's parameter type")
        if grads is not None and (grads.dtype != self.param_dtype or grads.numel() < self.nparams + 1):
            raise NativeError("grads must hold param_count+1 elements of the handle's parameter type")
        counts = [int(c) for c in batch_rows]
        if sum(counts) != x.shape[0] or any(c < 0 for c in counts):
            raise NativeError("batch_rows must be non-negative and sum to the number of rows of x")
        arr = (ctypes.c_int64 * len(counts))(*counts)
        hp = AdamHP(int(first_step), float(lr), float(beta1), float(beta2), float(eps))
        _check(lib().bamd_train_epoch_dp(self._h, _ptr(x), _dt(x), arr, len(counts), _ptr(features), _ptr(params), _ptr(grads),
                                         _ptr(m), _ptr(v), ctypes.byref(hp), _ptr(loss_accum), self._s()), "bamd_train_epoch_dp")
        return len(counts)

    def activation_means(self, x, features=None, max_nodes=200):
        x = _dev_tensor(x)
        self._mine(x, features)
        out = torch.empty((len(self.dims) - 3, max_nodes), dtype=torch.float64, device=x.device)
        _check(lib().bamd_activation_means(self._h, _ptr(x), _dt(x), x.shape[0], _ptr(features), _ptr(out),
                                           max_nodes, self._s()), "bamd_activation_means")
        return out
