"""Sliced-Wasserstein loss (utils.loss_function_swae) on the GPU: bamd_swd and the latent
gradient injection of bamd_fwd_bwd_latent against the torch restatement (oracle/torch_ref.py) and the
reference-generated fixture g15_swae.npz."""
import numpy as np
import pytest
import torch

from baler_amd import native, synth
from oracle import c_oracle as orc
from oracle import torch_ref

pytestmark = pytest.mark.gpu


def _swd_ref(z, prior, proj, reg_weight):
    z = torch.tensor(z, requires_grad=True)
    pm = torch.tensor(proj).transpose(0, 1)
    w = torch.sort(z.matmul(pm).t(), dim=1)[0] - torch.sort(torch.tensor(prior).matmul(pm).t(), dim=1)[0]
    loss = reg_weight * w.pow(2.0).mean()
    loss.backward()
    return float(loss.detach()), z.grad.numpy()


@pytest.mark.parametrize("n,d,s,dtype", [(64, 7, 2000, np.float32), (512, 15, 2000, np.float32), (37, 25, 100, np.float64),
                                         (4096, 6, 16, np.float32), (2, 3, 5, np.float64),
                                         # above 4096 rows: passes through global memory (CFD_project_animation_config.py:19: batch_size = 6000)
                                         (6000, 25, 12, np.float32), (4097, 7, 3, np.float64), (20000, 5, 4, np.float32)])
def test_swd_kernel(n, d, s, dtype):
    rng = np.random.default_rng(n + d)
    z = rng.normal(size=(n, d)).astype(dtype) * 0.3 + 0.1
    prior = rng.normal(size=(n, d)).astype(dtype)
    proj = rng.normal(size=(s, d))
    proj = (proj / np.linalg.norm(proj, axis=1, keepdims=True)).astype(dtype)
    rw = 100.0 / (n * (n - 1))
    want_loss, want_dz = _swd_ref(z.astype(np.float64), prior.astype(np.float64), proj.astype(np.float64), rw)
    loss, dz = native.swd(torch.from_numpy(z).cuda(), torch.from_numpy(prior).cuda(), torch.from_numpy(proj).cuda(), rw)
    tol = 1e-5 if dtype == np.float32 else 1e-12
    assert abs(loss.item() - want_loss) <= tol * want_loss
    # a float32 sort may order two nearly equal projections differently from float64: the gradient is continuous
    # across such swaps, so the norm-wise tolerance still holds
    assert np.linalg.norm(dz.cpu().numpy() - want_dz) <= 10 * tol * np.linalg.norm(want_dz)
    with pytest.raises(native.NativeError):
        native.swd(torch.zeros(1, 3).cuda(), torch.zeros(1, 3).cuda(), torch.zeros(4, 3).cuda(), 1.0)


def test_swae_step_golden(golden):
    """One loss_function_swae step on the fixture's float32 CFD_dense_AE(625, 7): encode -> bamd_swd ->
    bamd_fwd_bwd_latent reproduces the reference's loss, its mse / swd parts and the full gradient."""
    g = golden("g15_swae.npz")
    F_, Z, n = int(g["n_features"]), int(g["z_dim"]), int(g["n"])
    dims = orc.ae_dims(F_, Z)
    flat = orc.formula_params(dims, int(g["seed"]))
    x = torch.tensor(synth.cfd_field(int(g["frames"])).reshape(n, F_), dtype=torch.float32).cuda()
    h = native.Handle(dims, "fp32")
    p = torch.from_numpy(np.concatenate([flat, [0.0]]).astype(np.float32)).cuda()
    h.load_params(p)
    z = h.encode(x)
    assert np.abs(z.cpu().numpy()[:8] - g["z_head"]).max() < 1e-5 * np.abs(g["z_head"]).max()
    swd, dz = native.swd(z, torch.from_numpy(g["prior"]).cuda(), torch.from_numpy(g["proj"]).cuda(), 100.0 / (n * (n - 1)))
    grads = torch.zeros_like(p)
    h.fwd_bwd_latent(x, dz, grads)
    plain = torch.zeros_like(p)
    h.fwd_bwd(x, plain)
    assert abs(swd.item() - float(g["swd"])) < 1e-5 * float(g["swd"])
    assert abs(grads[-1].item() - float(g["mse"])) < 1e-5 * float(g["mse"])
    idx = g["grad_idx"]
    got = grads.cpu().numpy().astype(np.float64)[idx]
    assert np.abs(got - g["grad_sample"]).max() < 1e-5 * float(g["grad_l2"])
    # the regulariser's own share (1 % of the gradient norm here) on its own scale
    share = (grads - plain).cpu().numpy().astype(np.float64)[idx]
    assert np.linalg.norm(share - g["grad_swd_sample"]) < 1e-3 * np.linalg.norm(g["grad_swd_sample"])
    # no latent term -> identical to bamd_fwd_bwd, bit for bit, on the same (layer-wise) path
    again = torch.zeros_like(p)
    h.fwd_bwd_latent(x, torch.zeros_like(dz), again)
    assert torch.equal(again, plain)


def test_fwd_bwd_latent_on_fused_shape():
    """AE(24, 15) is served by the fused kernels; with a latent term the call runs layer-wise and must agree with
    torch autograd of mse + <c, z>."""
    dims = orc.ae_dims(24, 15)
    flat = orc.formula_params(dims, 5)
    xn = orc.normalize(synth.cms_rows(300))
    c = np.random.default_rng(1).normal(size=(300, 15)) * 0.01
    m = torch_ref.load_flat(torch_ref.DenseAE(24, 15), flat)
    xt = torch.tensor(xn)
    loss = torch.nn.functional.mse_loss(m(xt), xt, reduction="sum") / 24 + (m.encode(xt) * torch.tensor(c)).sum()
    loss.backward()
    want = torch.cat([q.grad.reshape(-1) for q in m.parameters()]).numpy()
    for mode, tol in (("fp32", 1e-5), ("fp64", 1e-11)):
        dt = torch.float32 if mode == "fp32" else torch.float64
        h = native.Handle(dims, mode)
        p = torch.from_numpy(np.concatenate([flat, [0.0]])).to(dt).cuda()
        h.load_params(p)
        grads = torch.zeros_like(p)
        h.fwd_bwd_latent(torch.from_numpy(xn).cuda(), torch.from_numpy(c).to(dt).cuda(), grads)
        got = grads.cpu().numpy().astype(np.float64)[:-1]
        assert np.linalg.norm(got - want) <= tol * np.linalg.norm(want)


_SWAE_CONFIG_EXTRA = '    c.custom_loss_function = "loss_function_swae"\n'


def test_cli_train_with_swae(tmp_path, monkeypatch):
    """--mode train with config.custom_loss_function = "loss_function_swae" (commented out in the reference's CFD /
    exafel configs): the run completes, the loss curve contains the regulariser, training reduces the loss."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    from test_gpu_cli import _CFD_CONFIG, _write_project
    cfg = (_CFD_CONFIG.replace("c.convert_to_blocks = False", "c.convert_to_blocks = [1, 25, 25]")
           .replace("c.batch_size = 6000", "c.batch_size = 64").replace("c.epochs = 3", "c.epochs = 4")) + _SWAE_CONFIG_EXTRA
    out = _write_project(tmp_path, monkeypatch, "CFD", "anim", cfg, synth.cfd_field(32), np.array([]))
    models.set_default_mode("fp32")
    init = orc.formula_params(orc.ae_dims(625, 7), 81)
    monkeypatch.setattr(helper, "model_init",
                        lambda name: (lambda n_features, z_dim: getattr(models, name)(n_features, z_dim).load_flat(init)))
    torch.manual_seed(0)
    baler.main(["--project", "CFD", "anim", "--mode", "train"])
    loss = np.load(out / "training" / "loss_data.npy")
    assert loss.shape == (2, 4) and np.all(np.isfinite(loss)) and loss[0][-1] < loss[0][0]
    # epoch 1 against the torch restatement driven by the same generator (same draws in the same order)
    x = torch.tensor(synth.cfd_field(32).reshape(128, 625), dtype=torch.float32)
    m = torch_ref.load_flat(torch_ref.DenseAE(625, 7, dtype=torch.float32), init)
    opt = torch.optim.Adam(m.parameters(), lr=1e-3)
    torch.manual_seed(0)
    tot = 0.0
    for s in range(0, 128, 64):
        xb = x[s:s + 64]
        with torch.no_grad():
            z = m.encode(xb)
        prior = torch.randn(z.shape, device="cuda").cpu()         # training draws randn_like(z) on the device
        proj = torch.randn(2000, 7)
        proj = proj / proj.norm(dim=1).view(-1, 1)
        l, _, _, g = torch_ref.swae_loss_and_grads(m, xb, prior, proj)
        opt.step()
        tot += l
    assert abs(loss[0][0] - tot / 2) < 1e-4 * (tot / 2)
