"""CPU-only tests: host logic of the product, the C-ABI library's exported symbols, and the
data-parallel sharding/all-reduce plumbing (world_size 2, gloo) with the oracle standing in for the
kernels.  No compute call crosses the C ABI here (there is no GPU in this container)."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from baler_amd import dist as bdist
from baler_amd import native, synth
from baler_amd.modules import helper, models, training, utils

from conftest import free_port

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(REPO, "include", "baler_amd.h")).read()
    declared = set(re.findall(r"\b(bamd_[a-z_0-9]+)\s*\(", header))
    declared -= {"bamd_handle", "bamd_status", "bamd_dtype", "bamd_mode", "bamd_adam"}
    lib = native.lib()
    for name in sorted(declared):
        assert hasattr(lib, name), name
    assert set(native.SYMBOLS) == declared
    assert lib.bamd_abi_version() == 1


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(native.NativeError):
        native.require_gpu()
    m = models.AE(24, 15)
    with pytest.raises(native.NativeError):
        m.encode(np.zeros((4, 24)))
    h = ctypes.c_void_p()
    dims = (ctypes.c_int * 9)(24, 200, 100, 50, 15, 50, 100, 200, 24)
    rc = native.lib().bamd_create(dims, 8, 0, 0, ctypes.byref(h))
    assert rc < 0 and native.lib().bamd_last_error()


def test_product_does_not_import_oracle():
    for root, _, files in os.walk(os.path.join(REPO, "baler_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")):
                src = open(os.path.join(root, f)).read()
                assert "oracle" not in src.replace("the oracle", ""), os.path.join(root, f)


def test_default_init_matches_torch_linear_init():
    from oracle import torch_ref
    torch.manual_seed(1234)
    m = models.AE(24, 15, mode="fp64")
    torch.manual_seed(1234)
    r = torch_ref.DenseAE(24, 15)
    assert np.array_equal(m.flat[:-1].numpy(), torch_ref.flat_of(r))
    assert m.nparams == 61839


def test_state_dict_roundtrip_reference_format(tmp_path):
    m = models.AE(24, 15, mode="fp64")
    sd = m.state_dict()
    keys = [f"{p}{i}.{k}" for p in ("en", "de") for i in (1, 2, 3, 4) for k in ("weight", "bias")]
    assert list(sd.keys()) == keys
    assert all(v.dtype == torch.float64 and v.device.type == "cpu" for v in sd.values())
    path = tmp_path / "model.pt"
    from baler_amd.modules import data_processing
    data_processing.save_model(m, str(path))
    m2 = models.AE(24, 15, mode="fp64")
    m2.load_state_dict(torch.load(str(path)), strict=False)
    assert torch.equal(m.flat, m2.flat)
    c = models.CFD_dense_AE(2500, 25)
    assert c.state_dict()["en1.weight"].dtype == torch.float32
    assert tuple(c.state_dict()["en1.weight"].shape) == (200, 2500)
    with pytest.raises(KeyError):
        m2.load_state_dict({}, strict=True)


def test_lr_scheduler_reference_sequence():
    # reference tests/test_utils.py:83-108 against the product's LRScheduler
    lin = torch.nn.Linear(10, 1)
    opt = torch.optim.SGD(lin.parameters(), lr=0.1)
    s = utils.LRScheduler(opt, patience=2, min_lr=1e-5, factor=0.5)
    for v in [10.0, 9.0, 8.0, 7.0]:
        s(v)
        assert opt.param_groups[0]["lr"] == 0.1
    for i, v in enumerate([10.0, 9.0, 10.0, 11.0, 12.0]):
        s(v)
        if i >= 2:
            assert opt.param_groups[0]["lr"] == 0.05
    for v in [10.0] * 100:
        s(v)
    assert opt.param_groups[0]["lr"] == 1e-5


def test_controllers_golden(golden):
    g = golden("g13_controllers.npz")

    class Opt:
        param_groups = [{"lr": 0.1}]

    o = Opt()
    s = utils.LRScheduler(o, patience=2)
    lrs = []
    for v in g["plateau_losses"]:
        s(v)
        lrs.append(o.param_groups[0]["lr"])
    assert lrs == list(g["plateau_lrs"])
    es = utils.EarlyStopping(3, 0.01)
    flags = []
    for v in g["es_losses"]:
        es(v)
        flags.append(es.early_stop)
    assert flags == list(g["es_flags"])
    from baler_amd.modules import data_processing
    tr, te = data_processing.split(np.arange(int(g["split_n"])), float(g["split_test_size"]), 1)
    assert np.array_equal(tr, g["split_train"]) and np.array_equal(te, g["split_test"])


def test_create_new_project_and_config(tmp_path, monkeypatch):
    # reference tests/test_helper.py:21-51
    monkeypatch.chdir(tmp_path)
    helper.create_new_project("ws", "proj", base_path="workspaces")
    for d in ("data", "proj/config", "proj/output/compressed_output", "proj/output/decompressed_output",
              "proj/output/plotting", "proj/output/training"):
        assert os.path.isdir(tmp_path / "workspaces" / "ws" / d)
    monkeypatch.syspath_prepend(str(tmp_path))
    cfg, mode, w, p, verbose = helper.get_arguments(["--project", "ws", "proj", "--mode", "train"])
    assert mode == "train" and cfg.model_name == "AE" and cfg.batch_size == 512 and cfg.lr == 0.001
    assert cfg.input_path == "workspaces/ws/data/proj_data.npz"
    cfg.Foo = "Bar"  # reference tests/test_data_processing.py:26-34: Config is a mutable holder
    assert helper.Config.Foo == "Bar"


def test_cms_config_keys():
    sys.path.insert(0, REPO)
    import importlib
    mod = importlib.import_module("workspaces.CMS_workspace.CMS_project_v1.config.CMS_project_v1_config")

    class C:
        pass

    mod.set_config(C)
    assert C.compression_ratio == 1.6 and C.batch_size == 512 and C.epochs == 25
    assert C.type_list == synth.CMS_TYPE_LIST


def test_batch_and_rank_slicing():
    spans = training._batches(10000, 512)
    assert len(spans) == 20 and spans[-1] == (9728, 10000)
    for lo, hi in [(0, 512), (9728, 10000), (0, 3), (5, 6)]:
        for world in (1, 2, 3, 8):
            parts = [training._rank_slice(lo, hi, r, world) for r in range(world)]
            assert parts[0][0] == lo and parts[-1][1] == hi
            assert all(parts[i][1] == parts[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
    assert bdist.shard_rows(10, 0, 3) == (0, 4) and bdist.shard_rows(10, 2, 3) == (7, 10)


def test_synth_generator_is_counter_based():
    a = synth.cms_rows(100, row0=0)
    b = synth.cms_rows(40, row0=60)
    assert np.array_equal(a[60:], b)
    ints = list(synth.CMS_INT_COLS)
    assert np.array_equal(a[:, ints], np.floor(a[:, ints])) and (a[:, ints] >= 0).all()
    assert (a[:, [c for c in range(24) if c not in ints]] > 0).all()


_DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
from baler_amd import dist as bdist, synth
from baler_amd.modules import training
from oracle import c_oracle as orc
rank, world, _ = bdist.init_from_env("gloo")
dims = orc.ae_dims(24, 15)
flat = orc.formula_params(dims, 3)
data = orc.normalize(synth.cms_rows(1100))
full_state = orc.FitState(dims, flat)
st = orc.FitState(dims, flat)
for lo, hi in training._batches(1100, 512):          # 512, 512, 76: ragged last global batch
    a, b = training._rank_slice(lo, hi, rank, world)
    if b > a:
        l, g = orc.fwd_bwd(dims, st.params, data[a:b])
    else:
        l, g = 0.0, np.zeros_like(st.params)
    buf = torch.from_numpy(np.concatenate([g, [l]]))
    bdist.allreduce_sum(buf)                           # the one exchange of the hot path: SUM
    g = buf.numpy()[:-1]
    st.t.value += 1
    orc.adam_step(st.params, g, st.m, st.v, st.t.value, 1e-3)
    lf, gf = orc.fwd_bwd(dims, full_state.params, data[lo:hi])
    full_state.t.value += 1
    orc.adam_step(full_state.params, gf, full_state.m, full_state.v, full_state.t.value, 1e-3)
    assert abs(buf.numpy()[-1] - lf) < 1e-12 * lf
    assert np.linalg.norm(g - gf) < 1e-12 * np.linalg.norm(gf)
assert np.linalg.norm(st.params - full_state.params) < 1e-12 * np.linalg.norm(full_state.params)
lo, hi = bdist.shard_rows(1100)
z = orc.encode(dims, flat, data[lo:hi])
parts = [None] * world
torch.distributed.all_gather_object(parts, z)
if rank == 0:
    assert np.array_equal(np.concatenate(parts), orc.encode(dims, flat, data))
# the sliced-Wasserstein exchange: ragged rank blocks (38 + 38, 39 + 38, 1 + 0 rows) gathered in rank order on every rank
assert not bdist.lib_comm_wanted()                     # gloo: the library-side RCCL step is never chosen
for lo, hi in ((1024, 1100), (0, 77), (5, 6)):
    spans = [training._rank_slice(lo, hi, r, world) for r in range(world)]
    counts = [b - a for a, b in spans]
    a, b = spans[rank]
    got = bdist.all_gather_rows(torch.from_numpy(data[a:b, :15].copy()), counts)
    assert got.shape == (hi - lo, 15) and np.array_equal(got.numpy(), data[lo:hi, :15]), (lo, hi, counts)
if rank == 0:
    print("DP-OK")
bdist.barrier()
'''


def test_data_parallel_world2_gloo(tmp_path):
    """DP step equivalence (sum of shard gradients == global-batch gradient, identical replicated Adam)
    and collective-free row sharding of compress, world_size 2 over gloo."""
    script = tmp_path / "dp_worker.py"
    script.write_text(_DP_WORKER)
    env = dict(os.environ, REPO=REPO, OMP_NUM_THREADS="1")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
         "--master-addr", "127.0.0.1", "--master-port", str(free_port()), str(script)],
        env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
    assert "DP-OK" in out.stdout


def test_forced_one_rank_process_group_gloo(tmp_path):
    """BALER_AMD_FORCE_PG=1: init_from_env builds a ONE-rank group and the data-parallel step keeps its collectives (what the
    GPU box runs over RCCL, tests/test_gpu_dp.py::test_rccl_world1_*); without the switch a single process has no group."""
    code = (
        "import os, sys\n"
        "sys.path.insert(0, os.environ['REPO'])\n"
        "import torch\n"
        "from baler_amd import dist as bdist\n"
        "assert not bdist.collectives_on()\n"
        "rank, world, local = bdist.init_from_env('gloo')\n"
        "forced = os.environ.get('BALER_AMD_FORCE_PG') == '1'\n"
        "assert (rank, world) == (0, 1) and bdist.is_dist() == forced and bdist.collectives_on() == forced\n"
        "t = torch.arange(5.0)\n"
        "assert torch.equal(bdist.allreduce_sum(t.clone()), t) and torch.equal(bdist.broadcast(t.clone()), t)\n"
        "bdist.barrier()\n"
        "print('PG-OK', forced)\n")
    for forced in ("1", "0"):
        env = dict(os.environ, REPO=REPO, BALER_AMD_FORCE_PG=forced, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(free_port()))
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert f"PG-OK {forced == '1'}" in out.stdout


def test_delta_side_channel_files_roundtrip(tmp_path):
    """split_deltas -> save_deltas -> load_deltas (host logic of helper.py:589-606, baler.py:316-338,
    helper.py:655-665), and the list-of-float16 form the reference's own writer produced loads the same way."""
    import gzip
    from baler_amd.modules import helper
    rng = np.random.default_rng(5)
    flags = (rng.uniform(size=(300, 24)) < 0.1).astype(np.uint8)
    flags[128:256] = 0                                      # a batch with nothing flagged
    deltas = rng.normal(size=(300, 24)).astype(np.float16)
    batches, dl, index = helper.split_deltas(flags, deltas, 128)
    assert batches == [0, 1, 2] and len(dl[1]) == 0 and len(index[1][0]) == 0
    helper.save_deltas(str(tmp_path), batches, dl, index)
    rows, cols, vals = helper.load_deltas(str(tmp_path / "compressed_deltas.npz.gz"),
                                          str(tmp_path / "compressed_batch_index_metadata.npz.gz"), 128)
    want_r, want_c = np.nonzero(flags)
    assert np.array_equal(rows, want_r) and np.array_equal(cols, want_c)
    assert vals.tobytes() == deltas[want_r, want_c].tobytes()
    # reference-style writer: Python lists of float16 scalars inside the object array
    d_arr = np.empty(3, dtype=object)
    for k in range(3):
        d_arr[k] = [np.float16(v) for v in dl[k]]
    with gzip.GzipFile(tmp_path / "compressed_deltas.npz.gz", "w") as f:
        np.save(file=f, arr=d_arr)
    rows2, cols2, vals2 = helper.load_deltas(str(tmp_path / "compressed_deltas.npz.gz"),
                                             str(tmp_path / "compressed_batch_index_metadata.npz.gz"), 128)
    assert np.array_equal(rows2, rows) and vals2.tobytes() == vals.tobytes()


def test_npz_array_shape_reads_header_only(tmp_path):
    from baler_amd.modules import helper
    a = np.arange(24.0).reshape(2, 3, 4)
    np.savez(tmp_path / "a.npz", data=a, names=np.array(["x"]))
    np.savez_compressed(tmp_path / "b.npz", data=a[0], names=np.array(["x"]))
    assert helper.npz_array_shape(str(tmp_path / "a.npz"), "data") == (2, 3, 4)
    assert helper.npz_array_shape(str(tmp_path / "b.npz"), "data") == (3, 4)
