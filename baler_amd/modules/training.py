"""Training loop (mirror of ``baler/modules/training.py``: fit / validate / train).

Same epoch semantics as the reference: the whole normalised dataset resident on the device
(training.py:230-231), sequential batches of ``batch_size`` rows, no shuffling, partial last batch
kept and weighted equally in the epoch mean (training.py:97-99), Adam with torch defaults
(training.py:266), ReduceLROnPlateau / EarlyStopping between epochs (training.py:269-323).

What differs is how a step executes: ONE native ``bamd_train_step`` per batch (forward + loss +
backward + Adam, fused HIP/MFMA kernels; data-parallel: ``bamd_fwd_bwd``, the gradient all-reduce,
then ``bamd_adam_step``), the running loss accumulated ON THE DEVICE and read once per epoch (the reference syncs with ``loss.item()`` every step,
training.py:97).  Under ``torch.distributed`` (one process per GPU, RCCL) every global batch is split
into contiguous row slices, one per rank, and the flat ``[grads | loss]`` buffer is SUM-all-reduced
between the two native calls; every rank then applies the identical Adam step.  A rank keeps ONLY its
slices resident (``ShardedRows``: the block-cyclic row shard of SURVEY.md section 8(e), built by
``helper.process``); a plain tensor / array is still accepted and sliced per batch.
"""
import os
import random
import sys
import time

import numpy as np
import torch

from .. import dist as bdist
from .. import hostio
from . import helper, utils


class Adam:
    """Flat-buffer Adam state (torch.optim.Adam defaults, reference training.py:266).  Exposes
    ``param_groups`` so the reference-style LRScheduler can drive ``lr``."""

    def __init__(self, model, lr, betas=(0.9, 0.999), eps=1e-8):
        self.model = model
        flat = model.flat
        self.param_groups = [{"lr": float(lr), "betas": betas, "eps": eps}]
        self.m = torch.zeros_like(flat)
        self.v = torch.zeros_like(flat)
        self.grads = torch.zeros_like(flat)       # [grads (nparams) | batch loss (1)]
        self.loss_accum = torch.zeros(1, dtype=torch.float64, device=flat.device)
        self.step_count = 0

    def zero_grad(self):
        """No work: bamd_fwd_bwd overwrites the gradient buffer (reference training.py:68)."""

    def train_step(self, handle, batch, world=1, swae_latent_dim=None, global_counts=None):
        """forward+loss+backward -> [all-reduce] -> Adam, all asynchronous on the current stream.  A single process
        issues ONE native call per batch (bamd_train_step: for small batches the Adam update is fused into the
        weight-gradient kernel); data-parallel ranks need the all-reduce between the two halves."""
        g = self.param_groups[0]
        if swae_latent_dim is not None:
            # loss = mse + sliced-Wasserstein(z) (training.py:73-80, utils.py:27-77): encode, regulariser forward +
            # backward on the latent batch, then the usual backward with dL/dz injected at the bottleneck
            z = handle.encode(batch, out_dtype=self.model.flat.dtype)
            if world > 1:
                # the regulariser sorts the latent codes of the WHOLE global batch: all-gather them (rank order = row order), draw
                # the prior sample and the projections once (rank 0) for everybody, evaluate the small kernel replicated, and inject
                # this rank's rows of dL/dz; the weight gradients then sum over the ranks like the reconstruction term's
                rank = bdist.rank_world()[0]
                counts = global_counts if global_counts is not None else [batch.shape[0]] * world
                n_glob = sum(counts)
                z_all = bdist.all_gather_rows(z, counts)
                draws = list(utils.swae_draws(z_all, swae_latent_dim, 2000, "normal"))
                for d in draws:
                    bdist.broadcast(d, src=0)
                swd, dz_all = utils.compute_swd(z_all, 2.0, 100 / (n_glob * (n_glob - 1)), swae_latent_dim, 2000, "normal", draws=tuple(draws))
                lo = sum(counts[:rank])
                if batch.shape[0] > 0:
                    handle.fwd_bwd_latent(batch, dz_all[lo:lo + batch.shape[0]].contiguous(), self.grads)
                else:
                    self.grads.zero_()          # no rows of this global batch here
                bdist.allreduce_sum(self.grads)
            else:
                reg_weight = 100 / (batch.shape[0] * (batch.shape[0] - 1))
                swd, dz = utils.compute_swd(z, 2.0, reg_weight, swae_latent_dim, 2000, "normal")
                handle.fwd_bwd_latent(batch, dz, self.grads)
            self.grads[-1:] += swd.to(self.grads.dtype)          # running loss = mse + swd, like loss.item()
            self.step_count += 1
            handle.adam_step(self.model.flat, self.grads, self.m, self.v, self.step_count, g["lr"], g["betas"][0],
                             g["betas"][1], g["eps"], loss_accum=self.loss_accum)
            return
        if (world == 1 and not bdist.collectives_on()) or handle.comm_world:
            # single process -- or a handle with its own RCCL communicator: the library runs fwd_bwd -> all-reduce -> Adam itself
            # (BALER_AMD_FORCE_PG=1 keeps the all-reduce at one rank)
            self.step_count += 1
            handle.train_step(batch, self.model.flat, self.m, self.v, self.step_count, g["lr"], g["betas"][0],
                              g["betas"][1], g["eps"], loss_accum=self.loss_accum, grads=self.grads)
            return
        handle.fwd_bwd(batch, self.grads)
        bdist.allreduce_sum(self.grads)
        self.step_count += 1
        handle.adam_step(self.model.flat, self.grads, self.m, self.v, self.step_count, g["lr"],
                         g["betas"][0], g["betas"][1], g["eps"], loss_accum=self.loss_accum)


def _epoch_in_one_call(world, swae_latent_dim):
    """A single process without the sliced-Wasserstein loss runs an epoch as ONE native call (bamd_train_epoch: the batch loop of
    training.py:64-97 inside the library; bit-identical to one bamd_train_step per batch).  BALER_AMD_EPOCH_CALL=0: per-step calls."""
    return (world == 1 and not bdist.collectives_on() and swae_latent_dim is None
            and os.environ.get("BALER_AMD_EPOCH_CALL", "1") != "0")


def _run_batches(optimizer, h, rows, spans, bs, world, swae_latent_dim, global_spans=None):
    """The optimiser steps of the batches `spans` (consecutive local row ranges of `rows`)."""
    if not spans:
        return
    if h.comm_world and swae_latent_dim is None and os.environ.get("BALER_AMD_EPOCH_CALL", "1") != "0":
        # data parallel with the communicator inside the library: the epoch's batch loop is ONE native call per rank
        a, b = spans[0][0], spans[-1][1]
        g = optimizer.param_groups[0]
        optimizer.step_count += h.train_epoch_dp(rows[a:b], [hi - lo for lo, hi in spans], optimizer.model.flat, optimizer.m,
                                                 optimizer.v, optimizer.step_count + 1, g["lr"], g["betas"][0], g["betas"][1],
                                                 g["eps"], loss_accum=optimizer.loss_accum, grads=optimizer.grads)
        return
    if _epoch_in_one_call(world, swae_latent_dim):
        a, b = spans[0][0], spans[-1][1]
        g = optimizer.param_groups[0]
        optimizer.step_count += h.train_epoch(rows[a:b], bs, optimizer.model.flat, optimizer.m, optimizer.v,
                                              optimizer.step_count + 1, g["lr"], g["betas"][0], g["betas"][1], g["eps"],
                                              loss_accum=optimizer.loss_accum, grads=optimizer.grads)
        return
    for i, (a, b) in enumerate(spans):
        counts = None
        if world > 1 and swae_latent_dim is not None:      # rows of every rank in this global batch (the latent all-gather)
            glo, ghi = global_spans[i]
            counts = [hi - lo for lo, hi in (_rank_slice(glo, ghi, r, world) for r in range(world))]
        optimizer.train_step(h, rows[a:b], world, swae_latent_dim=swae_latent_dim, global_counts=counts)


def _swae_dim(config, model):
    """Latent size when config.custom_loss_function selects the sliced-Wasserstein loss (training.py:73-76), else None."""
    if getattr(config, "custom_loss_function", None) == "loss_function_swae":
        return model.z_dim
    return None


def _batches(n_rows, bs):
    return [(s, min(s + bs, n_rows)) for s in range(0, n_rows, bs)]


def _rank_slice(lo, hi, rank, world):
    """Contiguous slice of global batch [lo,hi) owned by `rank` (sizes differ by at most one row)."""
    return hostio.RowPlan.slice_of(lo, hi, rank, world)


class ShardedRows:
    """The rows of a data-parallel dataset that THIS rank keeps resident: its slice of every global batch, stored
    back to back (``local``), with ``spans[i]`` = the local row range of global batch i.  ``shape`` is the GLOBAL
    shape, so callers that size the model from ``data.shape[1]`` (baler.py:107-125) see what they saw before."""

    def __init__(self, local, n_global, spans, global_batch, rank, world):
        self.local, self.n_global, self.spans = local, int(n_global), list(spans)
        self.global_batch, self.rank, self.world = int(global_batch), rank, world

    @property
    def shape(self):
        return (self.n_global,) + tuple(self.local.shape[1:])

    def map_local(self, fn):
        return ShardedRows(fn(self.local), self.n_global, self.spans, self.global_batch, self.rank, self.world)


def _local_batches(data, bs, rank, world):
    """-> (tensor to slice, [(lo, hi) of this rank's rows of every global batch]) for a ShardedRows or a replicated
    tensor (then the rank slices each global batch itself)."""
    if isinstance(data, ShardedRows):
        if data.global_batch != bs or data.world != world:
            raise ValueError(f"dataset was sharded for global batch {data.global_batch} on {data.world} ranks; "
                             f"the loop runs global batch {bs} on {world}")
        return data.local, data.spans
    spans = _batches(data.shape[0], bs)
    if world > 1:
        spans = [_rank_slice(lo, hi, rank, world) for lo, hi in spans]
    return data, spans


def fit(config, model, train_dl, model_children, regular_param, optimizer, latent_dim, RHO, l1,
        n_dimensions):
    """One epoch (reference training.py:31-101).  ``train_dl`` is a ``(device tensor, batch_size)``
    pair -- the native loop slices the resident tensor instead of going through a DataLoader.
    Returns (epoch_loss, last batch loss, 0, model) like the reference."""
    print("### Beginning Training")
    model.train()
    data, bs = train_dl
    rank, world = bdist.rank_world()
    h = model.handle()
    optimizer.loss_accum.zero_()
    rows, spans = _local_batches(data, bs, rank, world)
    _run_batches(optimizer, h, rows, spans, bs, world, _swae_dim(config, model), _batches(data.shape[0], bs))
    # one device->host read per epoch (the reference does one per step)
    last = float(optimizer.grads[model.nparams].item())
    epoch_loss = float(optimizer.loss_accum.item()) / len(spans)
    model._dirty = False  # the native Adam updated model.flat AND the handle's packed copy
    print(f"# Finished. Training Loss: {last:.6f}")
    return epoch_loss, last, 0, model


def validate(model, test_dl, model_children, reg_param):
    """reference training.py:104-137: mean over batches of the batch loss, no gradients."""
    print("### Beginning Validating")
    model.eval()
    data, bs = test_dl
    rank, world = bdist.rank_world()
    h = model.handle()
    rows, spans = _local_batches(data, bs, rank, world)
    losses = torch.zeros(len(spans), dtype=torch.float64, device=rows.device)
    for i, (a, b) in enumerate(spans):
        if b > a:
            h.forward_loss(rows[a:b], want_recon=False, loss_out=losses[i:i + 1])
    if world > 1:
        bdist.allreduce_sum(losses)
    host = losses.cpu().numpy()
    epoch_loss = float(host.sum() / len(spans))
    print(f"# Finished. Validation Loss: {host[-1]:.6f}")
    return epoch_loss


def _to_device_dataset(arr, config, device):
    """reference training.py:194-231: 1-D data float64 (n, c); 2-D dense data float32 (n, h*w)."""
    if isinstance(arr, ShardedRows):
        return arr.map_local(lambda t: _to_device_dataset(t, config, device))
    t = arr if isinstance(arr, torch.Tensor) else torch.as_tensor(np.asarray(arr))
    if config.data_dimension == 2:
        if getattr(config, "model_type", None) != "dense":
            raise NotImplementedError("baler_amd covers the dense models; convolutional models are out of scope")
        t = t.to(torch.float32).reshape(t.shape[0], -1)
    elif config.data_dimension == 1:
        t = t.to(torch.float64)
    return t.to(device).contiguous()


def train(model, variables, train_data, test_data, project_path, config):
    """reference training.py:150-348 (same artefacts: loss_data.npy, activations.npy, model_{epoch}.pt).
    ``train_data`` / ``test_data``: arrays / device tensors (every rank holds them whole and slices its share of a
    batch) or ``ShardedRows`` from ``helper.process`` (every rank holds only its share)."""
    if config.deterministic_algorithm:
        random.seed(0)
        torch.manual_seed(0)
        np.random.seed(0)
        # the native kernels are bitwise deterministic by construction (fixed-order reductions)

    test_size = config.test_size
    rank, world = bdist.rank_world()
    bs = bdist.global_batch(config, world)      # rows per optimiser step over all ranks (dist.batch_policy)
    epochs = config.epochs
    device = helper.get_device()
    model = model.to(device)
    model_children = list(model.children())
    # data parallel: the reference builds the model from an unseeded RNG (baler.py:158-160), so every rank holds
    # different initial weights -- rank 0's are broadcast once, after which the replicated Adam keeps ranks identical
    bdist.broadcast(model.flat, src=0)
    model.mark_params_changed()
    if bdist.lib_comm_wanted():      # RCCL: the handle gets its own communicator and runs the data-parallel step itself
        try:
            bdist.attach_comm(model.handle())
        except Exception as e:       # noqa: BLE001 -- e.g. a librccl the library cannot resolve: the three-call sequence still works
            print(f"[baler_amd] data-parallel step stays in Python (no library communicator: {type(e).__name__}: {e})", file=sys.stderr, flush=True)

    train_ds = _to_device_dataset(train_data, config, device)
    valid_ds = train_ds if test_data is train_data else _to_device_dataset(test_data, config, device)
    train_dl, valid_dl = (train_ds, bs), (valid_ds, bs)

    optimizer = Adam(model, lr=config.lr)
    if config.early_stopping:
        early_stopping = utils.EarlyStopping(patience=config.early_stopping_patience,
                                             min_delta=config.min_delta)
    if config.lr_scheduler:
        lr_scheduler = utils.LRScheduler(optimizer=optimizer, patience=config.lr_scheduler_patience)

    train_loss, val_loss = [], []
    start = time.time()
    want_acts = bool(getattr(config, "activation_extraction", False))
    trained_model = model

    for epoch in range(epochs):
        print(f"Epoch {epoch + 1} of {epochs}")
        # The reference's forward hooks fire on EVERY forward (models.py:160-183) and only the last one survives.
        # Without a validation split that is the epoch's last TRAINING batch with the weights it sees, i.e. before its
        # optimiser step: captured inside fit, in every epoch that can be the last (the final one; any, with early
        # stopping).  With a split it is the last VALIDATION batch of the last epoch with the post-step weights, which
        # are still the model's weights when the loop ends: captured once, after the loop.
        capture = want_acts and not test_size and (epoch == epochs - 1 or bool(config.early_stopping))
        train_epoch_loss, _, _, trained_model = _fit_with_capture(
            config, model, train_dl, model_children, optimizer, capture)
        train_loss.append(train_epoch_loss)

        if test_size:
            val_epoch_loss = validate(model=trained_model, test_dl=valid_dl, model_children=model_children,
                                      reg_param=config.reg_param)
        else:
            val_epoch_loss = train_epoch_loss
        val_loss.append(val_epoch_loss)

        if config.lr_scheduler:
            lr_scheduler(val_epoch_loss)
        if config.early_stopping:
            early_stopping(val_epoch_loss)
            if early_stopping.early_stop:
                break
        if config.intermittent_model_saving and rank == 0:
            if epoch % config.intermittent_saving_patience == 0:
                helper.model_saver(model, os.path.join(project_path, f"model_{epoch}.pt"))

    end = time.time()
    if want_acts and test_size:
        rows, spans = _local_batches(valid_ds, bs, rank, world)
        model._dirty = False
        _capture(model, rows, spans[-1], world)
    if rank == 0:
        if want_acts:
            acts = model.get_activations().get("means")
            if acts is not None:
                np.save(os.path.join(project_path, "activations.npy"), acts.cpu().numpy())
        print(f"{(end - start) / 60:.3} minutes")
        np.save(os.path.join(project_path, "loss_data.npy"), np.array([train_loss, val_loss]))
    return trained_model


def _capture(model, rows, span, world):
    """Activation means of one GLOBAL batch from this rank's rows of it: mean = sum over ranks of (local mean x
    local rows) / global rows -- one SUM all-reduce of the (6, 200) table plus the row count."""
    a, b = span
    if world == 1:
        model.capture_activations(rows[a:b])
        return
    if b > a:
        part = model.capture_activations(rows[a:b]) * float(b - a)
    else:
        part = model_nan_pattern(model, rows.device)      # no rows of this batch here: contributes 0 (NaN on the padding)
    cnt = torch.tensor([float(b - a)], dtype=torch.float64, device=rows.device)
    bdist.allreduce_sum(part)
    bdist.allreduce_sum(cnt)
    model.activations = {"means": part / cnt}


def model_nan_pattern(model, device):
    """(layers, 200) table that is NaN where activations.npy is padding (nodes beyond a layer's width), 0 elsewhere."""
    L = len(model.dims) - 1
    widths = [model.dims[l + 1] for l in range(L) if not (l == L // 2 - 1 or l == L - 1)]
    t = torch.zeros((len(widths), 200), dtype=torch.float64, device=device)
    for i, w in enumerate(widths):
        t[i, w:] = float("nan")
    return t


def _fit_with_capture(config, model, train_dl, model_children, optimizer, want_acts):
    """fit(), plus -- when asked -- the activation snapshot of the epoch's last batch taken with the weights that
    batch sees (i.e. before its optimiser step), as the reference's forward hooks do."""
    if not want_acts:
        return fit(config, model, train_dl, model_children, getattr(config, "reg_param", 0.0), optimizer,
                   getattr(config, "latent_space_size", None), getattr(config, "RHO", None),
                   getattr(config, "l1", None), config.data_dimension)
    data, bs = train_dl
    rank, world = bdist.rank_world()
    rows, spans = _local_batches(data, bs, rank, world)
    # all batches but the last, then capture, then the last batch
    print("### Beginning Training")
    model.train()
    h = model.handle()
    optimizer.loss_accum.zero_()
    gspans = _batches(data.shape[0], bs)
    _run_batches(optimizer, h, rows, spans[:-1], bs, world, _swae_dim(config, model), gspans[:-1])
    model._dirty = False      # the native Adam kept the handle's packed weights in step with model.flat
    _capture(model, rows, spans[-1], world)
    _run_batches(optimizer, h, rows, spans[-1:], bs, world, _swae_dim(config, model), gspans[-1:])
    last = float(optimizer.grads[model.nparams].item())
    epoch_loss = float(optimizer.loss_accum.item()) / len(spans)
    model._dirty = False
    print(f"# Finished. Training Loss: {last:.6f}")
    return epoch_loss, last, 0, model
