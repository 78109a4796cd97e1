"""End-to-end CLI on the GPU: train -> compress -> decompress on the C1 workload (10k x 24 synthetic
CMS rows, the reference's CMS config), compared with the artefacts of the reference CLI run recorded
in tests/golden/g7_c1_cli.npz."""
import os
import shutil
import sys

import numpy as np
import pytest
import torch

from baler_amd import synth
from oracle import c_oracle as orc

pytestmark = pytest.mark.gpu

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def rel(a, b):
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


@pytest.fixture()
def workspace(tmp_path, monkeypatch):
    _reset_config()     # helper.Config keeps attributes of earlier projects (e.g. custom_loss_function of the SWAE test)
    ws = tmp_path / "workspaces"
    shutil.copytree(os.path.join(REPO, "workspaces", "CMS_workspace"), ws / "CMS_workspace")
    (ws / "__init__.py").write_text("")
    for d in ("compressed_output", "decompressed_output", "plotting", "training"):
        os.makedirs(ws / "CMS_workspace" / "CMS_project_v1" / "output" / d, exist_ok=True)
    os.makedirs(ws / "CMS_workspace" / "data", exist_ok=True)
    np.savez(ws / "CMS_workspace" / "data" / "example_CMS_data.npz", data=synth.cms_rows(10000),
             names=synth.CMS_NAMES)
    monkeypatch.chdir(tmp_path)
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "workspaces" or k.startswith("workspaces.")]:
        del sys.modules[k]
    return ws / "CMS_workspace" / "CMS_project_v1" / "output"


def run_cli(mode_name, seed, compute_mode, monkeypatch):
    from baler_amd import baler
    from baler_amd.modules import helper, models
    models.set_default_mode(compute_mode)
    init = orc.formula_params(orc.ae_dims(24, 15), seed)

    def factory(name):
        cls = getattr(models, name)

        def make(n_features, z_dim):
            return cls(n_features, z_dim).load_flat(init)
        return make

    monkeypatch.setattr(helper, "model_init", factory)
    baler.main(["--project", "CMS_workspace", "CMS_project_v1", "--mode", mode_name])


@pytest.mark.parametrize("compute_mode", ["fp64", "fp32"])
def test_cli_train_compress_decompress(workspace, golden, monkeypatch, compute_mode):
    g = golden("g7_c1_cli.npz")
    out = workspace
    for mode_name in ("train", "compress", "decompress", "info"):
        run_cli(mode_name, int(g["init_seed"]), compute_mode, monkeypatch)

    loss = np.load(out / "training" / "loss_data.npy")
    assert loss.shape == (2, 25) and loss.dtype == np.float64
    assert np.array_equal(loss[0], loss[1])
    nf = np.load(out / "training" / "normalization_features.npy")
    assert np.array_equal(nf, g["normalization_features"])
    acts = np.load(out / "training" / "activations.npy")
    assert acts.shape == (6, 200) and np.array_equal(np.isnan(acts), np.isnan(g["activations"]))
    sd = torch.load(out / "compressed_output" / "model.pt")
    assert list(sd.keys())[0] == "en1.weight" and list(sd.keys())[-1] == "de4.bias"
    assert all(v.dtype == torch.float64 for v in sd.values())
    assert tuple(sd["en1.weight"].shape) == (200, 24) and tuple(sd["de4.bias"].shape) == (24,)
    comp = np.load(out / "compressed_output" / "compressed.npz")
    assert comp["data"].shape == (10000, 15) and comp["data"].dtype == np.float64
    assert np.array_equal(comp["names"], g["names"])
    assert np.array_equal(comp["normalization_features"], g["compressed_nf"])
    dec = np.load(out / "decompressed_output" / "decompressed.npz")
    assert dec["data"].shape == (10000, 24) and dec["data"].dtype == np.float64
    int_cols = [i for i, t in enumerate(synth.CMS_TYPE_LIST) if t == "int"]
    assert np.array_equal(dec["data"][:, int_cols], np.trunc(dec["data"][:, int_cols]))

    final = np.concatenate([v.numpy().ravel() for v in sd.values()])
    if compute_mode == "fp64":
        # fp64 device path pins the WHOLE run: 25 epochs x 20 steps
        assert rel(loss[0], g["loss_data"][0]) < 1e-9
        assert rel(final[g["final_sample_idx"]], g["final_sample"]) < 1e-6
        assert rel(np.nan_to_num(acts), np.nan_to_num(g["activations"])) < 1e-6
        assert rel(comp["data"][:64], g["compressed_head"]) < 1e-6
        assert rel(dec["data"].sum(axis=0), g["decompressed_colsum"]) < 1e-6
    else:
        # fp32 parity mode: 1e-5 while the trajectories are still numerically comparable, then loss-curve
        # agreement.  fp32-vs-fp64 training diverges chaotically; WHEN it leaves 1e-5 depends
        # on rounding details: measured epoch-4 deviation 2e-7 (throughput kernels) .. 6e-5 (latency kernel),
        # with per-step gradient errors of 1e-7 in both, so pin the first 60 steps
        assert rel(loss[0][:3], g["loss_data"][0][:3]) < 1e-5
        assert np.max(np.abs(loss[0] / g["loss_data"][0] - 1)) < 0.05
        # compress/decompress of this run's own model agree with the oracle at 1e-5
        data = orc.normalize(synth.cms_rows(10000))
        z = orc.encode(orc.ae_dims(24, 15), final, data)
        assert rel(comp["data"], z) < 1e-5


def test_cli_bf16_training_mode(workspace, golden, monkeypatch):
    """The whole C1 CLI run (25 epochs x 20 steps, then compress / decompress) with BALER_AMD_MODE=bf16: bf16 MFMA kernels
    for training AND inference, fp32 master weights.  SURVEY.md section 0 acceptance for a reduced-precision mode: the final
    loss agrees with the reference's (g7) within 5 % (the curve within 12 %); the checkpoint keeps the reference's format (float64 state dict)."""
    g = golden("g7_c1_cli.npz")
    out = workspace
    for mode_name in ("train", "compress", "decompress"):
        run_cli(mode_name, int(g["init_seed"]), "bf16", monkeypatch)
    loss = np.load(out / "training" / "loss_data.npy")
    assert loss.shape == (2, 25)
    dev = np.abs(loss[0] / g["loss_data"][0] - 1)
    # measured: 1e-4 .. 2e-3 over the first five epochs, up to 8 % where the two optimisation trajectories have separated
    # (epochs 20-23: the same chaotic divergence the fp32 mode shows against fp64), 2.7 % at the end
    assert dev[-1] < 0.05 and dev[:5].max() < 0.01 and dev.max() < 0.12
    sd = torch.load(out / "compressed_output" / "model.pt")
    assert all(v.dtype == torch.float64 for v in sd.values())
    final = np.concatenate([v.numpy().ravel() for v in sd.values()])
    comp = np.load(out / "compressed_output" / "compressed.npz")["data"]
    z = orc.encode(orc.ae_dims(24, 15), final, orc.normalize(synth.cms_rows(10000)))
    assert comp.shape == (10000, 15) and rel(comp, z) < 2e-2           # the bf16 inference bar
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    assert dec.shape == (10000, 24) and np.isfinite(dec).all()


def _reset_config():
    """helper.Config is a class mutated in place (like the reference's, helper.py:92-93): attributes of an
    earlier project (e.g. the CMS type_list) would leak into the next one within this test process."""
    from baler_amd.modules import helper
    for k in [k for k in vars(helper.Config) if not k.startswith("__")]:
        delattr(helper.Config, k)


def _write_project(tmp_path, monkeypatch, workspace, project, config_body, data, names):
    _reset_config()
    ws = tmp_path / "workspaces"
    proj = ws / workspace / project
    for d in ("config", "output/compressed_output", "output/decompressed_output", "output/plotting",
              "output/training"):
        os.makedirs(proj / d, exist_ok=True)
    os.makedirs(ws / workspace / "data", exist_ok=True)
    (ws / "__init__.py").write_text("")
    (proj / "config" / f"{project}_config.py").write_text(config_body)
    np.savez(ws / workspace / "data" / f"{project}.npz", data=data, names=names)
    monkeypatch.chdir(tmp_path)
    monkeypatch.syspath_prepend(str(tmp_path))
    for k in [k for k in sys.modules if k == "workspaces" or k.startswith("workspaces.")]:
        del sys.modules[k]
    return proj / "output"


_SPLIT_CONFIG = '''
def set_config(c):
    c.input_path = "workspaces/W/data/P.npz"
    c.data_dimension = 1
    c.compression_ratio = 1.6
    c.apply_normalization = True
    c.custom_norm = False
    c.model_name = "AE"
    c.epochs = 3
    c.lr = 0.001
    c.batch_size = 512
    c.test_size = 0.2
    c.early_stopping = True
    c.early_stopping_patience = 100
    c.min_delta = 0
    c.lr_scheduler = True
    c.lr_scheduler_patience = 50
    c.deterministic_algorithm = True
    c.reg_param = 0.001
    c.RHO = 0.05
    c.l1 = True
    c.activation_extraction = False
    c.intermittent_model_saving = True
    c.intermittent_saving_patience = 2
    c.separate_model_saving = False
    c.extra_compression = True
    c.save_error_bounded_deltas = False
    c.error_bounded_requirement = 10
    c.convert_to_blocks = False
'''


def test_cli_validation_split(tmp_path, monkeypatch):
    """test_size != 0: sklearn-style split (random_state=1), validate loop, intermittent saving,
    extra_compression -- against the CPU oracle replaying the same run."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    from oracle import host_logic
    raw = synth.cms_rows(6000, row0=20000)
    out = _write_project(tmp_path, monkeypatch, "W", "P", _SPLIT_CONFIG, raw, synth.CMS_NAMES)
    models.set_default_mode("fp64")
    init = orc.formula_params(orc.ae_dims(24, 15), 77)
    monkeypatch.setattr(helper, "model_init",
                        lambda name: (lambda n_features, z_dim: getattr(models, name)(n_features, z_dim).load_flat(init)))
    baler.main(["--project", "W", "P", "--mode", "train"])
    loss = np.load(out / "training" / "loss_data.npy")
    assert loss.shape == (2, 3)
    data = orc.normalize(raw)
    tr, te = host_logic.split_indices(6000, 0.2)
    st = orc.FitState(orc.ae_dims(24, 15), init)
    want = []
    for ep in range(3):
        el, _ = orc.fit_epoch(st, data[tr], 512, 1e-3)
        want.append((el, orc.validate_epoch(st.dims, st.params, data[te], 512)))
    assert rel(loss[0], [w[0] for w in want]) < 1e-9
    assert rel(loss[1], [w[1] for w in want]) < 1e-9
    assert os.path.exists(out / "training" / "model_0.pt") and os.path.exists(out / "training" / "model_2.pt")
    baler.main(["--project", "W", "P", "--mode", "compress"])
    baler.main(["--project", "W", "P", "--mode", "decompress"])
    comp = np.load(out / "compressed_output" / "compressed.npz")      # savez_compressed
    assert rel(comp["data"], orc.encode(st.dims, st.params, data)) < 1e-9
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    nf = orc.find_minmax(raw)
    assert rel(dec, orc.renormalize(orc.decode(st.dims, st.params, comp["data"]), nf[0], nf[1])) < 1e-9
    models.set_default_mode("fp32")


_CFD_CONFIG = '''
def set_config(c):
    c.input_path = "workspaces/CFD/data/anim.npz"
    c.compression_ratio = 100
    c.epochs = 3
    c.early_stopping = False
    c.early_stopping_patience = 100
    c.min_delta = 0
    c.lr_scheduler = True
    c.lr_scheduler_patience = 50
    c.model_name = "CFD_dense_AE"
    c.model_type = "dense"
    c.custom_norm = True
    c.l1 = True
    c.reg_param = 0.001
    c.RHO = 0.05
    c.lr = 0.001
    c.batch_size = 6000
    c.test_size = 0
    c.data_dimension = 2
    c.apply_normalization = False
    c.extra_compression = False
    c.intermittent_model_saving = False
    c.intermittent_saving_patience = 100
    c.activation_extraction = False
    c.deterministic_algorithm = False
    c.save_error_bounded_deltas = False
    c.error_bounded_requirement = 1
    c.convert_to_blocks = False
    c.separate_model_saving = False
'''


def test_cli_cfd_dense_2d(tmp_path, monkeypatch):
    """The reference's working CFD config (CFD_project_animation: CFD_dense_AE(2500,25), 2-D data flattened to
    (N, 2500) float32, one 60-row batch) end to end on the generic layer-wise MFMA path."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    field = synth.cfd_field(60)
    out = _write_project(tmp_path, monkeypatch, "CFD", "anim", _CFD_CONFIG, field, np.array([]))
    models.set_default_mode("fp32")
    dims = orc.ae_dims(2500, 25)
    init = orc.formula_params(dims, 78)
    monkeypatch.setattr(helper, "model_init",
                        lambda name: (lambda n_features, z_dim: getattr(models, name)(n_features, z_dim).load_flat(init)))
    for mode in ("train", "compress", "decompress"):
        baler.main(["--project", "CFD", "anim", "--mode", mode])
    loss = np.load(out / "training" / "loss_data.npy")
    x = field.astype(np.float32).astype(np.float64).reshape(60, 2500)
    st = orc.FitState(dims, init.astype(np.float32).astype(np.float64))
    want = [orc.fit_epoch(st, x, 6000, 1e-3)[0] for _ in range(3)]
    assert rel(loss[0], want) < 1e-4          # the reference model itself is float32 here
    sd = torch.load(out / "compressed_output" / "model.pt")
    assert sd["en1.weight"].dtype == torch.float32 and tuple(sd["en1.weight"].shape) == (200, 2500)
    comp = np.load(out / "compressed_output" / "compressed.npz")
    assert comp["data"].shape == (60, 25) and comp["data"].dtype == np.float32
    final = np.concatenate([v.numpy().ravel().astype(np.float64) for v in sd.values()])
    assert rel(comp["data"], orc.encode(dims, final, x)) < 1e-5
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    assert dec.shape == (60, 50, 50) and dec.dtype == np.float32
    assert rel(dec.reshape(60, 2500), orc.decode(dims, final, comp["data"].astype(np.float64))) < 1e-5


def test_cli_cfd_dense_2d_bf16_mode(tmp_path, monkeypatch):
    """The same CFD project with BALER_AMD_MODE=bf16: training runs the five wide products on the bf16 MFMA (fp32 master weights, fp32
    narrow layers and loss: the three-step loss curve still agrees to 1e-4 -- the loss of a pass differs by ~1e-6, its gradients by
    ~1e-3), compress / decompress run en1 / de4 on the bf16 MFMA (bf16-level agreement)."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    field = synth.cfd_field(60)
    out = _write_project(tmp_path, monkeypatch, "CFD", "anim", _CFD_CONFIG, field, np.array([]))
    models.set_default_mode("bf16")
    try:
        dims = orc.ae_dims(2500, 25)
        init = orc.formula_params(dims, 78)
        monkeypatch.setattr(helper, "model_init",
                            lambda name: (lambda n_features, z_dim: getattr(models, name)(n_features, z_dim).load_flat(init)))
        for mode in ("train", "compress", "decompress"):
            baler.main(["--project", "CFD", "anim", "--mode", mode])
    finally:
        models.set_default_mode("fp32")
    loss = np.load(out / "training" / "loss_data.npy")
    x = field.astype(np.float32).astype(np.float64).reshape(60, 2500)
    st = orc.FitState(dims, init.astype(np.float32).astype(np.float64))
    want = [orc.fit_epoch(st, x, 6000, 1e-3)[0] for _ in range(3)]
    assert rel(loss[0], want) < 1e-4
    sd = torch.load(out / "compressed_output" / "model.pt")
    comp = np.load(out / "compressed_output" / "compressed.npz")
    final = np.concatenate([v.numpy().ravel().astype(np.float64) for v in sd.values()])
    assert comp["data"].shape == (60, 25)
    assert rel(comp["data"], orc.encode(dims, final, x)) < 6e-3
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    assert dec.shape == (60, 50, 50)
    assert rel(dec.reshape(60, 2500), orc.decode(dims, final, comp["data"].astype(np.float64))) < 6e-3


@pytest.mark.parametrize("compute_mode", ["fp32", "bf16"])
def test_cli_cfd_dense_2d_class_shape(tmp_path, monkeypatch, compute_mode, capfd):
    """A 2-D field of a size that has NO exact instantiation (30 x 30 -> CFD_dense_AE(900, 9), models.py:192-209; three shipped configs
    feed un-blocked 2-D data of whatever size the file has) end to end on the run-time-width wide class: train (ragged last batch:
    45 + 45 + 10 frames) -> compress -> decompress against the oracle replaying the same run.  With BALER_AMD_MODE=bf16 the same project
    runs too -- in float32 with a notice (no bf16 kernels for this shape), same bars."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    field = synth.cfd_field(100, 30, 30)
    cfg = _CFD_CONFIG.replace("c.batch_size = 6000", "c.batch_size = 45").replace("c.epochs = 3", "c.epochs = 2")
    out = _write_project(tmp_path, monkeypatch, "CFD", "anim", cfg, field, np.array([]))
    models.set_default_mode(compute_mode)
    try:
        dims = orc.ae_dims(900, 9)
        init = orc.formula_params(dims, 80)
        monkeypatch.setattr(helper, "model_init",
                            lambda name: (lambda n_features, z_dim: getattr(models, name)(n_features, z_dim).load_flat(init)))
        for mode in ("train", "compress", "decompress"):
            baler.main(["--project", "CFD", "anim", "--mode", mode])
    finally:
        models.set_default_mode("fp32")
    if compute_mode == "bf16":
        assert "computes in float32" in capfd.readouterr().err
    loss = np.load(out / "training" / "loss_data.npy")
    x = field.astype(np.float32).astype(np.float64).reshape(100, 900)
    st = orc.FitState(dims, init.astype(np.float32).astype(np.float64))
    want = [orc.fit_epoch(st, x, 45, 1e-3)[0] for _ in range(2)]
    assert rel(loss[0], want) < 1e-4          # the reference model itself is float32 here
    sd = torch.load(out / "compressed_output" / "model.pt")
    assert tuple(sd["en1.weight"].shape) == (200, 900) and tuple(sd["en4.weight"].shape) == (9, 50)
    comp = np.load(out / "compressed_output" / "compressed.npz")
    assert comp["data"].shape == (100, 9) and comp["data"].dtype == np.float32
    final = np.concatenate([v.numpy().ravel().astype(np.float64) for v in sd.values()])
    assert rel(final, st.params) < 1e-3       # two epochs of Adam from the same start (float32 moves of 1e-3 per step)
    assert rel(comp["data"], orc.encode(dims, final, x)) < 1e-5
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    assert dec.shape == (100, 30, 30) and dec.dtype == np.float32
    assert rel(dec.reshape(100, 900), orc.decode(dims, final, comp["data"].astype(np.float64))) < 1e-5


def test_cli_blocks_2d_with_validation_split(tmp_path, monkeypatch):
    """The reference's exafel1/exafel2 shape of run (public_datasets/exafel1 config): 2-D frames cut into 25x25
    blocks (convert_to_blocks = [1, 25, 25], data_processing.py:26-34), CFD_dense_AE(625, 7) in float32,
    batch_size 32, test_size 0.2, no normalisation -- end to end against the CPU oracle replaying the same run
    (SURVEY section 8(f) row 4)."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    from oracle import host_logic
    field = synth.cfd_field(40)                                   # (40, 50, 50)
    cfg = (_CFD_CONFIG.replace("c.convert_to_blocks = False", "c.convert_to_blocks = [1, 25, 25]")
           .replace("c.batch_size = 6000", "c.batch_size = 32").replace("c.test_size = 0", "c.test_size = 0.2")
           .replace("c.epochs = 3", "c.epochs = 2"))
    out = _write_project(tmp_path, monkeypatch, "CFD", "anim", cfg, field, np.array([]))
    models.set_default_mode("fp32")
    dims = orc.ae_dims(625, 7)
    init = orc.formula_params(dims, 79)
    monkeypatch.setattr(helper, "model_init",
                        lambda name: (lambda n_features, z_dim: getattr(models, name)(n_features, z_dim).load_flat(init)))
    for mode in ("train", "compress", "decompress"):
        baler.main(["--project", "CFD", "anim", "--mode", mode])
    blocks = field.reshape(160, 25, 25)                           # total_size // (25*25) blocks, row-major
    x = blocks.astype(np.float32).astype(np.float64).reshape(160, 625)
    tr, te = host_logic.split_indices(160, 0.2)
    st = orc.FitState(dims, init.astype(np.float32).astype(np.float64))
    want = []
    for _ in range(2):
        el, _ = orc.fit_epoch(st, x[tr], 32, 1e-3)
        want.append((el, orc.validate_epoch(st.dims, st.params, x[te], 32)))
    loss = np.load(out / "training" / "loss_data.npy")
    assert loss.shape == (2, 2)
    assert rel(loss[0], [w[0] for w in want]) < 1e-4 and rel(loss[1], [w[1] for w in want]) < 1e-4
    sd = torch.load(out / "compressed_output" / "model.pt")
    assert tuple(sd["en1.weight"].shape) == (200, 625) and tuple(sd["en4.weight"].shape) == (7, 50)
    final = np.concatenate([v.numpy().ravel().astype(np.float64) for v in sd.values()])
    comp = np.load(out / "compressed_output" / "compressed.npz")
    assert comp["data"].shape == (160, 7)
    assert rel(comp["data"], orc.encode(dims, final, x)) < 1e-5
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    assert dec.shape == (40, 50, 50)                              # blocks folded back into the original frames
    assert rel(dec.reshape(160, 625), orc.decode(dims, final, comp["data"].astype(np.float64))) < 1e-5


def test_cli_compress_decompress_bf16_mode(workspace, golden, monkeypatch):
    """BALER_AMD_MODE=bf16 (the throughput mode of compress / decompress): same CLI, same artefacts and dtypes,
    outputs at the bf16 bar against the oracle's fp64 encode / decode of the same model."""
    from baler_amd import baler
    from baler_amd.modules import helper, models
    out = workspace
    _reset_config()
    flat = golden("g7_c1_model_f32.npz")["final_params_f32"].astype(np.float64)
    dims = orc.ae_dims(24, 15)
    raw = synth.cms_rows(10000)
    models.set_default_mode("bf16")
    try:
        helper.model_saver(models.AE(24, 15).load_flat(flat), str(out / "compressed_output" / "model.pt"))
        np.save(out / "training" / "normalization_features.npy", orc.find_minmax(raw))
        baler.main(["--project", "CMS_workspace", "CMS_project_v1", "--mode", "compress"])
        baler.main(["--project", "CMS_workspace", "CMS_project_v1", "--mode", "decompress"])
    finally:
        models.set_default_mode("fp32")
    comp = np.load(out / "compressed_output" / "compressed.npz")["data"]
    assert comp.dtype == np.float64 and comp.shape == (10000, 15)
    zo = orc.encode(dims, flat, orc.normalize(raw))
    assert rel(comp, zo) < 2e-2
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    nf = orc.find_minmax(raw)
    want = orc.renormalize(orc.decode(dims, flat, comp), nf[0], nf[1])
    fl = np.array([t != "int" for t in synth.CMS_TYPE_LIST])
    assert dec.dtype == np.float64 and rel(dec[:, fl], want[:, fl]) < 2e-2
    assert np.array_equal(dec[:, ~fl], np.trunc(dec[:, ~fl]))
