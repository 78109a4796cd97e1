#!/usr/bin/env python3
"""Generate tests/golden/g15_swae.npz by IMPORTING the reference (authoring container only).

Run:  python tools/gen_golden_swae.py       (needs /root/reference; writes tests/golden/)

Pins oracle/torch_ref.swae_loss_and_grads (the sliced-Wasserstein loss) against the
reference's ``utils.loss_function_swae`` + ``loss.backward()`` (utils.py:27-77, training.py:73-92) on a float32
``CFD_dense_AE(625, 7)`` (the loss only runs on float32 models in the reference: the projection matrix is float32
and ``z.matmul`` refuses mixed dtypes).  The reference draws ``prior_z`` and the projections from torch's global
generator; the generator script re-seeds and repeats the same two draws so the fixture can carry them.
"""
import os
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"
OUT = os.path.join(REPO, "tests", "golden")

import numpy as np
import torch

os.chdir(tempfile.mkdtemp(prefix="baler_golden_swae_"))
sys.path.insert(0, REF)
sys.path.insert(0, REPO)

from baler.modules import models as ref_models  # noqa: E402
from baler.modules import utils as ref_utils  # noqa: E402

from baler_amd import synth  # noqa: E402
from oracle import c_oracle as orc  # noqa: E402
from oracle import torch_ref  # noqa: E402

N, F, Z, S, SEED, TSEED = 64, 625, 7, 2000, 23, 1234


def main():
    torch.set_num_threads(4)
    dims = orc.ae_dims(F, Z)
    flat = orc.formula_params(dims, SEED)
    x = torch.tensor(synth.cfd_field(16).reshape(N, F), dtype=torch.float32)

    model = torch_ref.load_flat(ref_models.CFD_dense_AE(F, Z), flat)
    torch.manual_seed(TSEED)
    recon = model(x)
    z = model.encode(x)
    loss, mse, swd = ref_utils.loss_function_swae(x, z, recon, Z)
    loss.backward()
    g_ref = torch.cat([p.grad.reshape(-1) for p in model.parameters()]).numpy().astype(np.float64)

    # the same two draws, in the order compute_swd makes them (utils.py:59-66)
    torch.manual_seed(TSEED)
    prior = torch.randn_like(z)
    proj = ref_utils.get_random_projections("normal", Z, S)

    m2 = torch_ref.load_flat(torch_ref.DenseAE(F, Z, dtype=torch.float32), flat)
    o_loss, o_mse, o_swd, o_g = torch_ref.swae_loss_and_grads(m2, x, prior, proj)
    print("reference loss/mse/swd:", float(loss), float(mse), float(swd))
    print("restatement           :", o_loss, o_mse, o_swd)
    assert o_loss == float(loss) and o_mse == float(mse) and o_swd == float(swd)
    assert np.array_equal(o_g, g_ref)

    rng = np.random.default_rng(7)
    idx = np.sort(rng.choice(g_ref.size, size=1536, replace=False))
    # the regulariser's own share of the gradient (what bamd_swd + the latent injection must add)
    m3 = torch_ref.load_flat(torch_ref.DenseAE(F, Z, dtype=torch.float32), flat)
    m3.zero_grad()
    (torch.nn.functional.mse_loss(m3(x), x, reduction="sum") / F).backward()
    g_mse = torch.cat([p.grad.reshape(-1) for p in m3.parameters()]).numpy().astype(np.float64)
    path = os.path.join(OUT, "g15_swae.npz")
    np.savez(path, n=N, n_features=F, z_dim=Z, seed=SEED, frames=16, prior=prior.numpy(), proj=proj.numpy(),
             loss=float(loss), mse=float(mse), swd=float(swd), grad_idx=idx, grad_sample=g_ref[idx],
             grad_swd_sample=(g_ref - g_mse)[idx], grad_swd_l2=np.linalg.norm(g_ref - g_mse),
             grad_l2=np.linalg.norm(g_ref), grad_mse_l2=np.linalg.norm(g_mse), z_head=z.detach().numpy()[:8])
    print(f"wrote g15_swae.npz: {os.path.getsize(path) / 1024:.1f} KB; |g|={np.linalg.norm(g_ref):.4g} "
          f"|g_mse|={np.linalg.norm(g_mse):.4g} |g - g_mse|={np.linalg.norm(g_ref - g_mse):.4g}")


if __name__ == "__main__":
    main()
