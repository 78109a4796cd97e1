"""Dense autoencoder models with the reference's model protocol, backed by libbaler_amd.so.

Mirrors ``baler/modules/models.py`` for the two dense topologies on the hot path:

* ``AE``            (reference models.py:116-183) -- fp64 state dict, 24->200->100->50->z->...->24
* ``CFD_dense_AE``  (reference models.py:186-253) -- fp32 state dict, same topology, wide ends

Protocol kept: ctor ``(n_features, z_dim)``, ``encode / decode / forward``, ``state_dict`` /
``load_state_dict(strict=False)``, ``to``, ``train`` / ``eval``, ``children``, ``parameters``,
``store_hooks / get_activations / detach_hooks``.  The Linear layers are NOT ``torch.nn.Linear``:
parameters live in ONE flat device tensor in state-dict order and every forward/backward is a
hand-written HIP kernel behind the C ABI (``baler_amd/native.py``).  torch is the tensor container.

Compute mode (``fp32`` parity mode by default, ``fp64``, ``bf16``) comes from
``BALER_AMD_MODE`` or ``set_default_mode``; the state dict keeps the reference's dtype either way.
"""
import math
import os
from collections import OrderedDict

import numpy as np
import torch

from .. import native

_DEFAULT_MODE = os.environ.get("BALER_AMD_MODE", "fp32")
_LAYER_NAMES = ("en1", "en2", "en3", "en4", "de1", "de2", "de3", "de4")


def set_default_mode(mode):
    global _DEFAULT_MODE
    if mode not in native.MODE_NAMES:
        raise ValueError(f"unknown compute mode {mode!r}")
    _DEFAULT_MODE = mode


def ae_dims(n_features, z_dim):
    """Layer widths of the reference's dense AEs (models.py:128-136)."""
    return [int(n_features), 200, 100, 50, int(z_dim), 50, 100, 200, int(n_features)]


def tensor_layout(dims):
    """[(key, offset, shape)] of the flat parameter vector in state-dict order."""
    out, off = [], 0
    for l in range(len(dims) - 1):
        name = _LAYER_NAMES[l] if len(dims) - 1 == 8 else f"fc{l + 1}"
        shape = (dims[l + 1], dims[l])
        out.append((name + ".weight", off, shape))
        off += shape[0] * shape[1]
        out.append((name + ".bias", off, (dims[l + 1],)))
        off += dims[l + 1]
    return out, off


def default_init(dims, dtype):
    """torch.nn.Linear's default init drawn in the reference's construction order
    (kaiming_uniform_(a=sqrt(5)) for W then U(+-1/sqrt(fan_in)) for b, per layer; models.py:128-136),
    so ``torch.manual_seed(s)`` before construction gives the reference's initial weights."""
    parts = []
    for l in range(len(dims) - 1):
        w = torch.empty(dims[l + 1], dims[l], dtype=dtype)
        torch.nn.init.kaiming_uniform_(w, a=math.sqrt(5))
        bound = 1.0 / math.sqrt(dims[l])
        b = torch.empty(dims[l + 1], dtype=dtype)
        torch.nn.init.uniform_(b, -bound, bound)
        parts += [w.reshape(-1), b]
    return torch.cat(parts)


class _LayerView:
    """Stand-in for an nn.Linear child: exposes weight/bias views of the flat vector."""

    def __init__(self, model, index):
        self._m, self._i = model, index

    @property
    def weight(self):
        return self._m._tensor_view(2 * self._i)

    @property
    def bias(self):
        return self._m._tensor_view(2 * self._i + 1)


class DenseAE:
    state_dtype = torch.float64

    def __init__(self, n_features, z_dim, *args, mode=None, **kwargs):
        self.n_features = int(n_features)
        self.z_dim = int(z_dim)
        self.dims = ae_dims(n_features, z_dim)
        self.mode = mode or _DEFAULT_MODE
        self.layout, self.nparams = tensor_layout(self.dims)
        self.param_dtype = torch.float64 if self.mode in ("fp64", "f64") else torch.float32
        # master parameters: flat, state-dict order, +1 slot so the vector can double as a
        # [params | scratch] buffer; created on CPU like the reference, moved by .to()
        self.flat = torch.zeros(self.nparams + 1, dtype=self.param_dtype)
        self.flat[: self.nparams] = default_init(self.dims, self.state_dtype).to(self.param_dtype)
        self.training = True
        self.activations = {}
        self._hooks_on = False
        self._handle = None
        self._dirty = True

    # ---- torch.nn.Module-like surface ------------------------------------------------------------
    @property
    def device(self):
        return self.flat.device

    def to(self, device):
        device = torch.device(device)
        if device != self.flat.device:
            self.flat = self.flat.to(device)
            self._handle = None
            self._dirty = True
        return self

    def train(self, flag=True):
        self.training = flag
        return self

    def eval(self):
        return self.train(False)

    def children(self):
        return iter([_LayerView(self, i) for i in range(len(self.dims) - 1)])

    def parameters(self):
        return iter([self._tensor_view(i) for i in range(len(self.layout))])

    def _tensor_view(self, i):
        key, off, shape = self.layout[i]
        n = int(np.prod(shape))
        return self.flat[off:off + n].view(*shape)

    def state_dict(self):
        """Reference-format checkpoint: OrderedDict key -> CPU tensor of the reference dtype."""
        sd = OrderedDict()
        flat = self.flat.detach().to("cpu", self.state_dtype)
        for key, off, shape in self.layout:
            n = int(np.prod(shape))
            sd[key] = flat[off:off + n].clone().view(*shape)
        return sd

    def load_state_dict(self, sd, strict=True):
        missing = [k for k, _, _ in self.layout if k not in sd]
        if strict and missing:
            raise KeyError(f"missing keys in state_dict: {missing}")
        flat = self.flat.detach().to("cpu").clone()
        for key, off, shape in self.layout:
            if key in sd:
                t = torch.as_tensor(sd[key]).detach().to("cpu")
                if tuple(t.shape) != tuple(shape):
                    raise ValueError(f"size mismatch for {key}: {tuple(t.shape)} vs {tuple(shape)}")
                flat[off:off + t.numel()] = t.reshape(-1).to(self.param_dtype)
        self.flat = flat.to(self.flat.device)
        self._dirty = True
        return self

    def load_flat(self, vec):
        v = torch.as_tensor(np.asarray(vec)).to(self.param_dtype)
        self.flat[: self.nparams] = v.to(self.flat.device)
        self._dirty = True
        return self

    @property
    def type(self):
        return f"{self.__class__.__name__}(dims={self.dims}, mode={self.mode})"

    # ---- native handle ---------------------------------------------------------------------------
    def handle(self):
        """bamd handle on the model's device with the packed weights in sync with ``flat``."""
        if not self.flat.is_cuda:
            raise native.NativeError(
                "model is on the CPU: the baler_amd hot path runs only on an MI355X (no CPU fallback); "
                "call model.to('cuda:0') on a GPU box")
        if self._handle is None:
            with torch.cuda.device(self.flat.device):
                self._handle = native.Handle(self.dims, self.mode, self.flat.device.index)
            self._dirty = True
        if self._dirty:
            with torch.cuda.device(self.flat.device):
                self._handle.load_params(self.flat)
            self._dirty = False
        return self._handle

    def mark_params_changed(self):
        self._dirty = True

    def _as_input(self, x):
        if isinstance(x, np.ndarray):
            x = torch.from_numpy(np.ascontiguousarray(x))
        if x.dtype not in (torch.float32, torch.float64):
            x = x.to(torch.float64)
        if x.dim() > 2:
            x = x.reshape(x.shape[0], -1)
        return x.to(self.flat.device).contiguous()

    def encode(self, x):
        x = self._as_input(x)
        return self.handle().encode(x)

    def decode(self, z):
        z = self._as_input(z)
        return self.handle().decode(z)

    def forward(self, x):
        x = self._as_input(x)
        if self._hooks_on:
            self._last_hook_input = x
        recon, _ = self.handle().forward_loss(x)
        return recon

    __call__ = forward

    # ---- activation extraction (reference models.py:160-183) -------------------------------------
    def get_layers(self):
        kids = list(self.children())
        L = len(kids)
        return [kids[l] for l in range(L) if not (l == L // 2 - 1 or l == L - 1)]

    def store_hooks(self):
        self._hooks_on = True
        self._last_hook_input = None
        return ["bamd-activation-capture"]

    def capture_activations(self, x, features=None):
        """Per-node mean of leaky_relu(pre-activation) over the rows of x for every activated layer
        -- what the reference's hooks + diagnostics.dict_to_square_matrix produce for the last batch."""
        x = self._as_input(x)
        self.activations = {"means": self.handle().activation_means(x, features)}
        return self.activations["means"]

    def get_activations(self):
        if not self.activations and getattr(self, "_last_hook_input", None) is not None:
            self.capture_activations(self._last_hook_input)
        return self.activations

    def detach_hooks(self, hooks):
        self._hooks_on = False


class AE(DenseAE):
    """reference models.AE (models.py:116-183): float64 checkpoint."""
    state_dtype = torch.float64


class CFD_dense_AE(DenseAE):
    """reference models.CFD_dense_AE (models.py:186-253): float32 checkpoint."""
    state_dtype = torch.float32
