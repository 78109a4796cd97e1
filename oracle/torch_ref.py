"""Plain-PyTorch fp64 restatement of the reference's hot loops -- TEST INFRASTRUCTURE ONLY.

This is what ``bench.py``'s ``cpu_baseline`` leg times on the GPU box's host cores (the
reference's own Python files never travel).  It executes the same ATen CPU kernels the reference
executes (addmm / leaky_relu / mse_loss(sum) / autograd / optim.Adam single-tensor) in the same
order, and is checked bit-for-bit against the imported reference in the authoring container by
``tools/gen_golden.py`` (fixtures ``tests/golden/ae24_train.npz``).

Reference loops restated: training.fit (training.py:31-101), training.train's tensor/DataLoader
construction (training.py:194-266), helper.compress's encode loop (helper.py:564-611),
helper.decompress's decode loop (helper.py:691-723).
"""
import numpy as np
import torch
from torch import nn
from torch.nn import functional as F
from torch.utils.data import DataLoader


class DenseAE(nn.Module):
    """8-layer dense AE with the reference's topology (models.py:116-156, 186-229)."""

    def __init__(self, n_features, z_dim, dtype=torch.float64):
        super().__init__()
        self.en1 = nn.Linear(n_features, 200, dtype=dtype)
        self.en2 = nn.Linear(200, 100, dtype=dtype)
        self.en3 = nn.Linear(100, 50, dtype=dtype)
        self.en4 = nn.Linear(50, z_dim, dtype=dtype)
        self.de1 = nn.Linear(z_dim, 50, dtype=dtype)
        self.de2 = nn.Linear(50, 100, dtype=dtype)
        self.de3 = nn.Linear(100, 200, dtype=dtype)
        self.de4 = nn.Linear(200, n_features, dtype=dtype)

    def encode(self, x):
        h = F.leaky_relu(self.en1(x))
        h = F.leaky_relu(self.en2(h))
        h = F.leaky_relu(self.en3(h))
        return self.en4(h)

    def decode(self, z):
        h = F.leaky_relu(self.de1(z))
        h = F.leaky_relu(self.de2(h))
        h = F.leaky_relu(self.de3(h))
        return self.de4(h)

    def forward(self, x):
        return self.decode(self.encode(x))


def load_flat(model, flat):
    """Load a flat fp64 vector in state-dict order into the model."""
    off = 0
    sd = model.state_dict()
    for k, v in sd.items():
        n = v.numel()
        sd[k] = torch.as_tensor(np.asarray(flat[off:off + n]).reshape(tuple(v.shape)), dtype=v.dtype)
        off += n
    model.load_state_dict(sd)
    return model


def flat_of(model):
    return np.concatenate([v.detach().cpu().double().numpy().ravel() for v in model.state_dict().values()])


def batch_loss(recon, x):
    """utils.mse_sum_loss_l1(validate=True) (utils.py:195-211)."""
    return nn.MSELoss(reduction="sum")(recon, x) / x.shape[1]


def fit_epoch(model, optimizer, train_dl):
    """training.fit (training.py:59-99): returns (epoch_loss, last batch loss)."""
    model.train()
    running = 0.0
    idx = -1
    loss = None
    for idx, inputs in enumerate(train_dl):
        optimizer.zero_grad()
        recon = model(inputs)
        loss = batch_loss(recon, inputs)
        loss.backward()
        optimizer.step()
        running += loss.item()
    return running / (idx + 1), float(loss.item())


def make_loader(data, bs, dtype=torch.float64):
    """training.py:230-263: whole dataset as one tensor, sequential batches, partial batch kept."""
    ds = torch.tensor(np.asarray(data), dtype=dtype)
    return DataLoader(ds, batch_size=bs, shuffle=False, drop_last=False)


def train_epochs(model, data, bs, lr, epochs):
    """Epoch loop without controllers; returns list of epoch losses."""
    dl = make_loader(data, bs)
    opt = torch.optim.Adam(model.parameters(), lr=lr)
    return [fit_epoch(model, opt, dl)[0] for _ in range(epochs)]


def compress_loop(model, data, bs):
    """helper.compress's loop (helper.py:564-611) incl. per-batch .numpy() and growing concatenate."""
    model.eval()
    dl = DataLoader(torch.tensor(np.asarray(data), dtype=torch.float64), batch_size=bs,
                    shuffle=False, drop_last=False)
    compressed = []
    with torch.no_grad():
        for idx, batch in enumerate(dl):
            out = model.encode(batch).cpu().detach().numpy()
            compressed = out if idx == 0 else np.concatenate((compressed, out))
    return compressed


def decompress_loop(model, z, bs):
    """helper.decompress's loop (helper.py:691-723)."""
    model.eval()
    dl = DataLoader(torch.from_numpy(np.asarray(z)), batch_size=bs, shuffle=False, drop_last=False)
    out_all = []
    with torch.no_grad():
        for idx, batch in enumerate(dl):
            out = model.decode(batch).cpu().detach().numpy()
            out_all = out if idx == 0 else np.concatenate((out_all, out))
    return out_all


def swae_loss_and_grads(model, x, prior_z, proj, reg_weight=100.0):
    """utils.loss_function_swae + loss.backward() (utils.py:27-77, training.py:73-92) with the two random draws
    (prior_z = randn_like(z); proj = unit rows [S x D]) passed in instead of taken from torch's global generator.
    Returns (loss, mse_sum_loss, swd, flat gradient in state-dict order).  Pinned against the reference by
    tools/gen_golden_swae.py (tests/golden/g15_swae.npz)."""
    model.zero_grad()
    recon = model(x)
    z = model.encode(x)
    bsz = x.shape[0]
    rw = reg_weight / (bsz * (bsz - 1))
    mse = F.mse_loss(recon, x, reduction="sum") / x.shape[1]
    pm = proj.transpose(0, 1)
    w = torch.sort(z.matmul(pm).t(), dim=1)[0] - torch.sort(prior_z.matmul(pm).t(), dim=1)[0]
    swd = rw * w.pow(2.0).mean()
    loss = mse + swd
    loss.backward()
    g = torch.cat([p.grad.reshape(-1) for p in model.parameters()])
    return float(loss.detach()), float(mse.detach()), float(swd.detach()), g.detach().numpy().astype(np.float64)
