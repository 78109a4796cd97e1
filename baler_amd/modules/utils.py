"""Loss helpers and per-epoch controllers (mirror of ``baler/modules/utils.py`` for the hot path).

* ``mse_sum_loss_l1``  -- reference utils.py:176-211; only ``validate=True`` is ever used by the
  reference's training loop (training.py:83-89), and that is what is implemented.
* ``emd_rows``         -- the EMD term of reference utils.py:112-119 as a device kernel.
* ``EarlyStopping``    -- reference utils.py:248-282.
* ``LRScheduler``      -- reference utils.py:285-323 (= ReduceLROnPlateau(mode="min", factor, patience,
  min_lr) with torch defaults threshold=1e-4 'rel', cooldown=0, eps=1e-8), written out on the host so
  it has no dependency on a torch optimizer; it accepts any object with ``param_groups``.
"""
import math

import torch

from .. import native

factor = 0.5
min_lr = 1e-6


def mse_sum_loss_l1(model_children, true_data, reconstructed_data, reg_param, validate):
    """sum((recon - x)^2) / n_columns on device; returns (loss, 0, 0) like the reference."""
    if not validate:
        raise NotImplementedError(
            "mse_sum_loss_l1(validate=False) is never reached from the reference CLI (training.py:83-89)")
    x = true_data.reshape(true_data.shape[0], -1).contiguous()
    r = reconstructed_data.reshape(x.shape).contiguous()
    d = (r - x).to(torch.float64)
    return (d * d).sum() / x.shape[1], 0, 0


def get_random_projections(proj_dist, latent_dim, num_samples):
    """reference utils.py:79-91: num_samples random unit vectors of the latent space, drawn from torch's global
    CPU generator like the reference does.  [S x D]"""
    if proj_dist == "normal":
        rand_samples = torch.randn(num_samples, latent_dim)
    elif proj_dist == "cauchy":
        rand_samples = (torch.distributions.Cauchy(torch.tensor([0.0]), torch.tensor([1.0]))
                        .sample((num_samples, latent_dim)).squeeze())
    else:
        raise ValueError("Unknown projection distribution.")
    return rand_samples / rand_samples.norm(dim=1).view(-1, 1)


def swae_draws(z, latent_dim, num_projections=2000, projection_dist="normal"):
    """The two random tensors of one loss_function_swae call, in the reference's order (utils.py:59-66):
    prior_z = randn_like(z), then the projection matrix.  -> (prior (n, d), proj (S, d)), on z's device and dtype."""
    prior_z = torch.randn_like(z)
    proj = get_random_projections(projection_dist, latent_dim, num_projections).to(device=z.device, dtype=z.dtype)
    return prior_z, proj.contiguous()


def compute_swd(z, p, reg_weight, latent_dim, num_projections, proj_dist, draws=None):
    """reference utils.py:58-77 on the native kernel (bamd_swd): -> (swd loss tensor (1,), dL/dz)."""
    if float(p) != 2.0:
        raise NotImplementedError("the sliced-Wasserstein kernel implements wasserstein_deg = 2 (the reference default)")
    prior_z, proj = draws if draws is not None else swae_draws(z, latent_dim, num_projections, proj_dist)
    return native.swd(z.contiguous(), prior_z.contiguous(), proj, reg_weight)


def loss_function_swae(inputs, z, reconstructions, latent_dim, reg_weight=100, wasserstein_deg=2.0,
                       num_projections=2000, projection_dist="normal", draws=None):
    """reference utils.py:27-55: (loss, mse_sum_loss, SWD) as device tensors (forward values; the training loop takes
    the gradient from bamd_swd + bamd_fwd_bwd_latent)."""
    batch_size = inputs.shape[0]
    reg_weight = reg_weight / (batch_size * (batch_size - 1))
    mse_sum_loss, _, _ = mse_sum_loss_l1(None, inputs, reconstructions, 0, True)
    swd, _ = compute_swd(z, wasserstein_deg, reg_weight, latent_dim, num_projections, projection_dist, draws)
    return mse_sum_loss + swd[0], mse_sum_loss, swd[0]


def mse_loss_emd_l1(model_children, true_data, reconstructed_data, reg_param, validate):
    """reference utils.py:94-132 with ``validate=True``: the summed per-row 1-D Wasserstein distance (a float).  The
    training variant is unreachable in the reference (and adds to an empty tensor, utils.py:122-128)."""
    if not validate:
        raise NotImplementedError("mse_loss_emd_l1(validate=False) is dead code in the reference (utils.py:122-130)")
    return float(emd_rows(true_data, reconstructed_data))


def emd_rows(true_data, reconstructed_data):
    """Sum over rows of the 1-D Wasserstein distance between a row's columns (device kernel)."""
    return native.emd_rows(true_data.contiguous(), reconstructed_data.contiguous())


class EarlyStopping:
    """Stops a run whose epoch loss has stopped improving (reference: utils.py:240-282; behaviour pinned by fixture g13).

    The state is the best loss seen and a strike counter.  Per epoch the GAIN over the best loss decides: a gain above
    ``min_delta`` is a new best and clears the strikes; a gain below it is a strike, and ``patience`` strikes set
    ``early_stop``.  A gain of exactly ``min_delta`` (or a NaN loss) is neither -- the reference tests both sides strictly.
    Attribute names and the two progress lines are the reference's (callers and logs read them)."""

    def __init__(self, patience: int, min_delta: float):
        self.patience, self.min_delta = patience, min_delta
        self.counter, self.best_loss, self.early_stop = 0, None, False

    def _strike(self):
        self.counter += 1
        print(f"Early stopping counter {self.counter} of {self.patience}")
        if self.counter >= self.patience:
            print("Early Stopping")
            self.early_stop = True

    def __call__(self, train_loss):
        if self.best_loss is None:              # first epoch: nothing to compare with
            self.best_loss = train_loss
            return
        gain = self.best_loss - train_loss
        if gain > self.min_delta:
            self.best_loss, self.counter = train_loss, 0
        elif gain < self.min_delta:
            self._strike()


class LRScheduler:
    def __init__(self, optimizer, patience, min_lr=min_lr, factor=factor):
        self.optimizer = optimizer
        self.patience = patience
        self.min_lr = min_lr
        self.factor = factor
        self.threshold = 1e-4
        self.eps = 1e-8
        self.best = math.inf
        self.num_bad_epochs = 0

    def __call__(self, train_loss):
        current = float(train_loss)
        if current < self.best * (1.0 - self.threshold):
            self.best = current
            self.num_bad_epochs = 0
        else:
            self.num_bad_epochs += 1
        if self.num_bad_epochs > self.patience:
            for group in self.optimizer.param_groups:
                old = float(group["lr"])
                new = max(old * self.factor, self.min_lr)
                if old - new > self.eps:
                    group["lr"] = new
            self.num_bad_epochs = 0
