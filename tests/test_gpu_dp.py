"""Multi-rank code path on the GPU box.  RCCL needs one GPU per rank and gpurun boxes have one GPU, so these
tests run TWO ranks on the same GPU over gloo (BALER_AMD_FORCE_DEVICE / BALER_AMD_DIST_BACKEND test hooks):
everything except the transport -- sharding, the [grads|loss] sum-all-reduce between bamd_fwd_bwd and
bamd_adam_step, replicated Adam, rank-0 artefacts, bench.py's barrier/max-over-ranks timing -- is the code the
8-GPU run executes."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import free_port

pytestmark = pytest.mark.gpu
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
from baler_amd import dist as bdist, synth
from baler_amd.modules import models, training, helper
from oracle import c_oracle as orc
rank, world, local = bdist.init_from_env()
torch.cuda.set_device(local)
dims = orc.ae_dims(24, 15)
init = orc.formula_params(dims, 5)
data = orc.normalize(synth.cms_rows(3000))

class Cfg: pass
c = Cfg()
c.deterministic_algorithm = False; c.test_size = 0; c.batch_size = 512; c.epochs = 3; c.lr = 1e-3
c.early_stopping = False; c.lr_scheduler = True; c.lr_scheduler_patience = 50; c.reg_param = 0.001
c.data_dimension = 1; c.activation_extraction = False; c.intermittent_model_saving = False
c.intermittent_saving_patience = 100
out = os.environ["OUT"] + f"/rank{rank}"
os.makedirs(out, exist_ok=True)
model = models.AE(24, 15, mode=os.environ.get("MODE", "fp64"))
if rank == 0:
    model.load_flat(init)   # only rank 0 holds the intended initial weights: train() must broadcast them
training.train(model, 24, data, data, out, c)
flat = model.flat.cpu().numpy().astype(np.float64)[:-1]
np.save(os.environ["OUT"] + f"/params_rank{rank}.npy", flat)
bdist.barrier()
'''


def _torchrun(nproc, script_or_args, env, port):
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}",
           "--master-addr", "127.0.0.1", "--master-port", str(port)] + script_or_args
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)


def test_dp_training_two_ranks_one_gpu(tmp_path):
    from baler_amd import synth
    from oracle import c_oracle as orc
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    env = dict(os.environ, REPO=REPO, OUT=str(tmp_path), BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo",
               MODE="fp64")
    r = _torchrun(2, [str(script)], env, free_port())
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    p0, p1 = np.load(tmp_path / "params_rank0.npy"), np.load(tmp_path / "params_rank1.npy")
    assert np.array_equal(p0, p1)                       # replicated Adam: identical on every rank
    loss = np.load(tmp_path / "rank0" / "loss_data.npy")
    assert not os.path.exists(tmp_path / "rank1" / "loss_data.npy")   # rank 0 writes the artefacts
    # == the single-process run with the same GLOBAL batch size (the oracle)
    dims = orc.ae_dims(24, 15)
    st = orc.FitState(dims, orc.formula_params(dims, 5))
    data = orc.normalize(synth.cms_rows(3000))
    want = [orc.fit_epoch(st, data, 512, 1e-3)[0] for _ in range(3)]
    assert np.linalg.norm(loss[0] - want) / np.linalg.norm(want) < 1e-9
    assert np.linalg.norm(p0 - st.params) / np.linalg.norm(st.params) < 1e-8


def test_bench_two_ranks_one_gpu(tmp_path):
    env = dict(os.environ, BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo")
    r = _torchrun(2, [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "65536",
                      "--no-extras"], env, free_port())
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["frac"] > 0


def test_bench_scale_record_two_ranks_driver_launch():
    """The round-end SCALE run, rehearsed: the driver's own launch line (`python -m torch.distributed.run --nnodes=1
    --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N --steps K --warmup W`, default extras) with two
    ranks on the one GPU, BALER_AMD_DP_BATCH=per_gpu in the environment as a per-GPU-batch run would have it.  Asserts what the
    SCALE record is computed from -- ONE JSON line from rank 0 with `value`, `rccl_ranks`, `allreduce_us`, both batch policies at
    batch_size 512, bit-identical replicas, the configs[2] leg -- and that every rank wrote its diagnostic lines (device, XCD
    count, IPC mode) BEFORE the first collective, so that a failed multi-GPU run can be read from its stderr tail."""
    env = dict(os.environ, BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo", BALER_AMD_DP_BATCH="per_gpu",
               HSA_ENABLE_IPC_MODE_LEGACY="0")
    r = _torchrun(2, [os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--rows", "65536",
                      "--c3-rows", "65536", "--pcie-rows", "200000", "--cpu-rows", "20000"], env, free_port())
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                "dtype", "data", "config", "roofline", "rccl_ranks", "allreduce_us", "allreduce_bytes", "allreduce_frac_of_step",
                "train_rows_per_s_by_batch_policy", "train_rows_per_s_by_batch", "train_rows_per_s_by_global_batch",
                "replicas_identical", "c3", "c3_train_rows_per_s"):
        assert key in d, key
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["scaling"] == "weak" and d["replicas_identical"] is True
    assert d["allreduce_us"] > 0 and d["train_rows_per_s_by_batch_policy"] == "per_gpu"
    assert d["train_rows_per_s_by_batch"]["512"]["rows_per_s"] > 0          # 512 rows PER GPU per step (global batch 1024)
    assert d["train_rows_per_s_by_global_batch"]["512"]["rows_per_gpu_per_step"] == 256
    assert d["c3"]["global_rows"] == 131072 and d["c3"]["train_rows_per_s"] > 0
    for rank in (0, 1):
        tag = f"rank {rank}/2"
        mine = [l for l in r.stderr.splitlines() if tag in l]
        assert any("start" in l and "HSA_ENABLE_IPC_MODE_LEGACY=0" in l for l in mine), r.stderr[-3000:]
        assert any("process group up" in l and "XCDs=8" in l and "CUs=256" in l for l in mine), r.stderr[-3000:]
        assert any("first all-reduce OK" in l for l in mine), r.stderr[-3000:]
    # the diagnostics come before the first collective of the run
    err = r.stderr
    assert err.index("process group up") < err.index("first all-reduce OK") < err.index("data resident")


_CLI_WORKER = r'''
import os, sys, shutil
sys.path.insert(0, os.environ["REPO"])
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from baler_amd import baler, dist as bdist, synth
from baler_amd.modules import helper, models, training
from oracle import c_oracle as orc
rank, world, local = bdist.init_from_env()
torch.cuda.set_device(local)
models.set_default_mode(os.environ.get("MODE", "fp64"))
init = orc.formula_params(orc.ae_dims(24, 15), 5)

def factory(name):
    cls = getattr(models, name)
    def make(n_features, z_dim):
        m = cls(n_features, z_dim)
        return m.load_flat(init) if rank == 0 else m       # rank 0's weights are broadcast by training.train
    return make
helper.model_init = factory

# what a rank keeps resident: ONLY its slice of every global batch, normalised with the whole table's min/max
cfg, *_ = helper.get_arguments(["--project", "CMS_workspace", "CMS_project_v1", "--mode", "train"])
G = bdist.global_batch(cfg, world)
tr, te, feats, shape = helper.process(cfg.input_path, cfg.custom_norm, cfg.test_size, cfg.apply_normalization, None, False,
                                      batch_size=G)
raw = synth.cms_rows(3000)
full = orc.normalize(raw)
assert isinstance(tr, training.ShardedRows) and tr is te and tr.shape == (3000, 24) and tuple(shape) == (3000, 24)
mine = np.concatenate([full[slice(*training._rank_slice(lo, min(lo + G, 3000), rank, world))] for lo in range(0, 3000, G)])
assert tr.local.shape[0] == len(mine) and abs(len(mine) - 3000 / world) <= 6
assert np.array_equal(tr.local.cpu().numpy(), mine)                      # bit-identical to the single-process table
assert np.array_equal(feats, np.stack([raw.min(0), raw.max(0) - raw.min(0)]))
np.save(os.environ["OUT"] + f"/resident_rows_rank{rank}.npy", np.array([tr.local.shape[0]]))

for mode in ("train", "compress", "decompress"):
    baler.main(["--project", "CMS_workspace", "CMS_project_v1", "--mode", mode])
bdist.barrier()
'''


def _dp_workspace(tmp_path, epochs=3, extra=""):
    from baler_amd import synth
    ws = tmp_path / "workspaces"
    proj = ws / "CMS_workspace" / "CMS_project_v1"
    for d in ("config", "output/compressed_output", "output/decompressed_output", "output/plotting", "output/training"):
        os.makedirs(proj / d, exist_ok=True)
    os.makedirs(ws / "CMS_workspace" / "data", exist_ok=True)
    (ws / "__init__.py").write_text("")
    src = open(os.path.join(REPO, "workspaces", "CMS_workspace", "CMS_project_v1", "config", "CMS_project_v1_config.py")).read()
    assert "c.epochs = 25" in src
    src = src.replace("c.epochs = 25", f"c.epochs = {epochs}")
    (proj / "config" / "CMS_project_v1_config.py").write_text(src + extra)
    np.savez(ws / "CMS_workspace" / "data" / "example_CMS_data.npz", data=synth.cms_rows(3000), names=synth.CMS_NAMES)
    return proj / "output"


@pytest.mark.parametrize("policy", ["global", "per_gpu"])
def test_cli_dp_sharded_residency(tmp_path, policy):
    """train -> compress -> decompress through the real CLI on 2 ranks (one GPU, gloo): every rank reads, uploads and
    keeps only its rows; the artefacts equal the single-process oracle run with the same GLOBAL batch."""
    from baler_amd import synth
    from oracle import c_oracle as orc
    out = _dp_workspace(tmp_path)
    script = tmp_path / "cli_worker.py"
    script.write_text(_CLI_WORKER)
    env = dict(os.environ, REPO=REPO, OUT=str(tmp_path), BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo",
               MODE="fp64", BALER_AMD_DP_BATCH=policy)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
           "--master-port", str(free_port()), str(script)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(tmp_path))
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    rows = [int(np.load(tmp_path / f"resident_rows_rank{k}.npy")[0]) for k in range(2)]
    assert sum(rows) == 3000 and max(rows) <= 1506                        # half the table per rank, not all of it
    G = 512 if policy == "global" else 1024
    dims = orc.ae_dims(24, 15)
    st = orc.FitState(dims, orc.formula_params(dims, 5))
    raw = synth.cms_rows(3000)
    data = orc.normalize(raw)
    want = [orc.fit_epoch(st, data, G, 1e-3)[0] for _ in range(3)]
    loss = np.load(out / "training" / "loss_data.npy")
    assert np.linalg.norm(loss[0] - want) / np.linalg.norm(want) < 1e-9
    comp = np.load(out / "compressed_output" / "compressed.npz")["data"]
    z = orc.encode(dims, st.params, data)
    assert comp.shape == (3000, 15) and np.linalg.norm(comp - z) / np.linalg.norm(z) < 1e-8
    dec = np.load(out / "decompressed_output" / "decompressed.npz")["data"]
    rec = orc.decode(dims, st.params, z) * (raw.max(0) - raw.min(0)) + raw.min(0)
    int_cols = [i for i, t in enumerate(synth.CMS_TYPE_LIST) if t == "int"]
    rec[:, int_cols] = np.trunc(rec[:, int_cols])
    assert dec.shape == (3000, 24)
    close = np.isclose(dec, rec, rtol=1e-7, atol=1e-9)
    assert close.mean() > 0.9999                                          # a truncation may flip where rec sits on an integer


def test_bench_spawns_two_ranks_one_gpu():
    """`python bench.py --gpus 2` as the driver runs it (no torchrun, no WORLD_SIZE): the parent starts the two ranks
    itself and relays the single JSON line."""
    env = dict(os.environ, BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
                        "--rows", "65536", "--c3-rows", "131072"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["rccl_ranks"] == 2 and d["dist_backend"] == "gloo" and d["value"] > 0
    # first multi-GPU contact: the all-reduce alone, bit-identical replicas, both batch policies, the configs[2] leg
    assert d["replicas_identical"] is True and d["allreduce_us"] > 0 and d["allreduce_bytes"] == 4 * 61840
    assert d["train_rows_per_s_by_batch_policy"] == "per_gpu" and "512" in d["train_rows_per_s_by_batch"]
    assert d["train_rows_per_s_by_global_batch"]["512"]["rows_per_gpu_per_step"] == 256
    c3 = d["c3"]
    assert c3["rows_per_gpu"] == 131072 and c3["global_rows"] == 262144 and c3["train_rows_per_s"] > 0
    assert 0 < c3["roofline"]["frac"] < 1 and d["c3_train_rows_per_s"] == c3["train_rows_per_s"]


_SPLIT_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
sys.path.insert(0, os.getcwd())
import numpy as np, torch
from baler_amd import baler, dist as bdist
from baler_amd.modules import helper, models
from oracle import c_oracle as orc
rank, world, local = bdist.init_from_env()
torch.cuda.set_device(local)
models.set_default_mode("fp64")
init = orc.formula_params(orc.ae_dims(24, 15), 5)
def factory(name):
    cls = getattr(models, name)
    return lambda n_features, z_dim: cls(n_features, z_dim).load_flat(init)
helper.model_init = factory
baler.main(["--project", "CMS_workspace", "CMS_project_v1", "--mode", "train"])
'''


def test_dp_validation_split_equals_single_process(tmp_path):
    """test_size = 0.2 + activation extraction + early stopping off, 4 epochs, fp64: the 2-rank run (index-sharded train and
    validation sets, per-batch validation losses and activation means combined by all-reduce) writes the SAME loss_data.npy,
    activations.npy and model as the single-process run of the same config (sums are re-associated across ranks: 1e-10)."""
    outs = {}
    for world in (1, 2):
        base = tmp_path / f"w{world}"
        os.makedirs(base)
        out = _dp_workspace(base, epochs=4, extra="\n_orig = set_config\ndef set_config(c):\n    _orig(c)\n    c.test_size = 0.2\n"
                                                  "    c.early_stopping = False\n    c.batch_size = 500\n")
        script = base / "w.py"
        script.write_text(_SPLIT_WORKER)
        env = dict(os.environ, REPO=REPO, BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
            env.pop(k, None)
        if world == 1:
            cmd = [sys.executable, str(script)]
        else:
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
                   "--master-port", str(free_port()), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(base))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        import torch
        sd = torch.load(out / "compressed_output" / "model.pt")
        outs[world] = (np.load(out / "training" / "loss_data.npy"), np.load(out / "training" / "activations.npy"),
                       np.concatenate([v.numpy().ravel() for v in sd.values()]))
    l1, a1, p1 = outs[1]
    l2, a2, p2 = outs[2]
    assert l1.shape == (2, 4) and not np.array_equal(l1[0], l1[1])           # a real validation loss
    assert np.allclose(l2, l1, rtol=1e-10, atol=0)
    assert np.array_equal(np.isnan(a1), np.isnan(a2)) and np.allclose(np.nan_to_num(a2), np.nan_to_num(a1), rtol=1e-9, atol=1e-12)
    assert np.linalg.norm(p2 - p1) / np.linalg.norm(p1) < 1e-10


# ---- RCCL itself, on the one GPU of the box: a ONE-rank "nccl" process group (BALER_AMD_FORCE_PG=1) -----------------------
# Every multi-rank test above runs over gloo (RCCL needs one GPU per rank).  A world-1 RCCL group is legal on one GPU and runs
# everything of the data-parallel step except the wire: init_process_group("nccl", device_id=...), RCCL's library load and
# communicator, dmabuf IPC mode, and the stream hand-off between RCCL's stream and the launch stream around bamd_fwd_bwd /
# bamd_adam_step (a sum over one rank is the identity, so any mis-ordering shows as a changed bit).

def _bench_json(r):
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def test_rccl_world1_bench_step_is_bit_identical():
    args = [os.path.join(REPO, "bench.py"), "--gpus", "1", "--steps", "3", "--warmup", "2", "--rows", "65536", "--no-extras",
            "--no-cpu-baseline"]
    env = dict(os.environ, BALER_AMD_FORCE_PG="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    env.pop("BALER_AMD_DIST_BACKEND", None)
    env.pop("BALER_AMD_FORCE_DEVICE", None)
    # the child is started before anything in IT touches the GPU (torchrun forks the rank); nothing is re-exec'd
    d = _bench_json(_torchrun(1, args, env, free_port()))
    assert d["n_gpus"] == 1 and d["rccl_ranks"] == 1 and d["dist_backend"] == "nccl"
    assert d["allreduce_us"] > 0 and d["allreduce_bytes"] == 4 * 61840 and d["value"] > 0
    # the same steps without any process group: gradients and parameters must agree to the last bit
    env0 = dict(os.environ)
    for k in ("BALER_AMD_FORCE_PG", "WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "MASTER_ADDR", "BALER_AMD_DIST_BACKEND"):
        env0.pop(k, None)
    r0 = subprocess.run([sys.executable] + args, env=env0, capture_output=True, text=True, timeout=900)
    d0 = _bench_json(r0)
    assert d0["dist_backend"] is None and "allreduce_us" not in d0
    assert d["param_checksum"] == d0["param_checksum"] and d["grad_checksum"] == d0["grad_checksum"]
    assert d["last_batch_loss"] == d0["last_batch_loss"]
    assert "RCCL version" in r0.stderr          # printed on every run: bench.py --gpus 1 names the library it would use


def test_rccl_world1_cli_train_is_bit_identical(tmp_path):
    """`python -m baler_amd --mode train` under a one-rank RCCL group (fwd_bwd -> ncclAllReduce -> adam_step per batch) writes the
    same loss curve and model as the plain single-process run (one fused bamd_train_step per batch): with the communicator INSIDE
    the library (default on RCCL: one bamd_train_epoch_dp call per epoch, `[baler_amd] data-parallel step inside the library` on
    stderr) and with the three-call Python sequence (BALER_AMD_LIB_COMM=0)."""
    import torch
    outs = {}
    for tag in ("plain", "rccl", "rccl-python"):
        base = tmp_path / tag
        os.makedirs(base)
        out = _dp_workspace(base, epochs=3)
        script = base / "w.py"
        script.write_text(_SPLIT_WORKER)
        env = dict(os.environ, REPO=REPO, HSA_ENABLE_IPC_MODE_LEGACY="0")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BALER_AMD_DIST_BACKEND", "BALER_AMD_FORCE_DEVICE", "BALER_AMD_FORCE_PG"):
            env.pop(k, None)
        env.pop("BALER_AMD_LIB_COMM", None)
        if tag == "plain":
            cmd = [sys.executable, str(script)]
        else:
            env["BALER_AMD_FORCE_PG"] = "1"
            if tag == "rccl-python":
                env["BALER_AMD_LIB_COMM"] = "0"
            cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr", "127.0.0.1",
                   "--master-port", str(free_port()), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900, cwd=str(base))
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
        sd = torch.load(out / "compressed_output" / "model.pt")
        outs[tag] = (np.load(out / "training" / "loss_data.npy"), np.concatenate([v.numpy().ravel() for v in sd.values()]))
        outs[tag] += (r.stderr,)
    assert outs["plain"][0].shape == (2, 3)
    for tag in ("rccl", "rccl-python"):
        assert np.array_equal(outs["plain"][0], outs[tag][0]), tag    # bamd_train_step == bamd_fwd_bwd + identity + bamd_adam_step
        assert np.array_equal(outs["plain"][1], outs[tag][1]), tag
    assert "data-parallel step inside the library" in outs["rccl"][2]
    assert "data-parallel step inside the library" not in outs["rccl-python"][2]


_LIBCOMM_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
from baler_amd import native, synth
from oracle import c_oracle as orc
mode = os.environ["MODE"]
dt = torch.float64 if mode == "fp64" else torch.float32
dims = orc.ae_dims(24, 15)
p0 = torch.from_numpy(np.concatenate([orc.formula_params(dims, 5), [0.0]])).to(dt).cuda()
x = torch.from_numpy(orc.normalize(synth.cms_rows(2000))).cuda()
counts = [512, 0, 300, 512, 64, 1, 611]                  # ragged, with an empty slice (a rank without rows of a batch)
res = []
for dp in (False, True):
    h = native.Handle(dims, mode)
    p = p0.clone(); h.load_params(p)
    if dp:
        h.comm_init(native.comm_unique_id(), 0, 1)       # RCCL communicator of ONE rank, created inside the library
        assert h.comm_world == 1
    m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
    la = torch.zeros(1, dtype=torch.float64, device="cuda")
    if dp:
        h.train_epoch_dp(x, counts, p, m, v, 1, 1e-3, loss_accum=la, grads=g)
        t = torch.arange(8, dtype=dt, device="cuda")
        assert torch.equal(h.allreduce_sum(t.clone()), t)
        h.comm_release()
        assert h.comm_world == 0
    else:
        r0 = 0
        for i, c in enumerate(counts):
            h.fwd_bwd(x[r0:r0 + c], g)
            h.adam_step(p, g, m, v, i + 1, 1e-3, loss_accum=la)
            r0 += c
    torch.cuda.synchronize()
    res.append((p.clone(), m.clone(), v.clone(), g.clone(), la.item()))
for a, b in zip(res[0], res[1]):
    assert (torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b)
# bamd_comm_attach: a communicator the CALLER made, with the RCCL the library itself resolved (the copy mapped in this process)
import ctypes
paths = sorted({l.split()[-1] for l in open("/proc/self/maps") if "librccl" in l})
assert len(paths) == 1, paths
rccl = ctypes.CDLL(paths[0])
class Uid(ctypes.Structure):
    _fields_ = [("internal", ctypes.c_char * 128)]
uid = Uid()
assert rccl.ncclGetUniqueId(ctypes.byref(uid)) == 0
comm = ctypes.c_void_p()
rccl.ncclCommInitRank.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, Uid, ctypes.c_int]
assert rccl.ncclCommInitRank(ctypes.byref(comm), 1, uid, 0) == 0 and comm.value
h = native.Handle(dims, mode)
p = p0.clone(); h.load_params(p)
h.comm_attach(comm.value)                                # world <= 0: the library asks ncclCommCount
assert h.comm_world == 1
m, v, g = torch.zeros_like(p), torch.zeros_like(p), torch.zeros_like(p)
la = torch.zeros(1, dtype=torch.float64, device="cuda")
h.train_epoch_dp(x, counts, p, m, v, 1, 1e-3, loss_accum=la, grads=g)
torch.cuda.synchronize()
for a, b in zip(res[0], (p, m, v, g, la.item())):
    assert (torch.equal(a, b) if isinstance(a, torch.Tensor) else a == b)
h.comm_release()
assert h.comm_world == 0
n = ctypes.c_int(-1)                                     # the handle did not destroy what it did not make
rccl.ncclCommCount.argtypes = [ctypes.c_void_p, ctypes.POINTER(ctypes.c_int)]
assert rccl.ncclCommCount(comm, ctypes.byref(n)) == 0 and n.value == 1
rccl.ncclCommDestroy.argtypes = [ctypes.c_void_p]
assert rccl.ncclCommDestroy(comm) == 0
h.fwd_bwd(x[:512], g)                                    # and it trains single-process again
torch.cuda.synchronize()
print("LIBCOMM_OK", res[0][4])
'''


@pytest.mark.parametrize("mode", ["fp32", "fp64", "bf16"])
def test_lib_comm_world1_epoch_equals_the_three_call_sequence(tmp_path, mode):
    """bamd_comm_unique_id / bamd_comm_init / bamd_comm_attach / bamd_train_epoch_dp / bamd_allreduce_sum / bamd_comm_release on RCCL with ONE rank
    (no torch.distributed at all: the library resolves and drives RCCL itself): the data-parallel epoch in one host call is
    bit-identical to bamd_fwd_bwd + bamd_adam_step per batch, ragged batch sizes and an empty slice included."""
    script = tmp_path / "w.py"
    script.write_text(_LIBCOMM_WORKER)
    env = dict(os.environ, REPO=REPO, MODE=mode, HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BALER_AMD_DIST_BACKEND", "BALER_AMD_FORCE_DEVICE", "BALER_AMD_FORCE_PG"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(script)], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "LIBCOMM_OK" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


_SWAE_DP_WORKER = r'''
import os, sys
sys.path.insert(0, os.environ["REPO"])
import numpy as np, torch
from baler_amd import dist as bdist, synth
from baler_amd.modules import models, training
from oracle import c_oracle as orc
rank, world, local = bdist.init_from_env()
torch.cuda.set_device(local)
class Cfg: pass
c = Cfg()
c.deterministic_algorithm = False; c.test_size = 0; c.batch_size = 64; c.epochs = 2; c.lr = 1e-3
c.early_stopping = False; c.lr_scheduler = False; c.reg_param = 0.001; c.data_dimension = 1
c.activation_extraction = False; c.intermittent_model_saving = False; c.intermittent_saving_patience = 100
c.custom_loss_function = "loss_function_swae"
data = orc.normalize(synth.cms_rows(151))                # batches of 64, 64, 23 rows: 32+32, 32+32, 12+11 per rank
model = models.AE(24, 15, mode="fp64").load_flat(orc.formula_params(orc.ae_dims(24, 15), 5))
out = os.environ["OUT"] + f"/w{world}r{rank}"
os.makedirs(out, exist_ok=True)
torch.manual_seed(1234 + 77 * rank)                       # rank 0 draws what the single process draws; rank 1's draws are never used
training.train(model, 24, data, data, out, c)
np.save(os.environ["OUT"] + f"/swae_params_w{world}r{rank}.npy", model.flat.cpu().numpy())
bdist.barrier()
'''


def test_dp_swae_equals_single_process(tmp_path):
    """config.custom_loss_function = "loss_function_swae" under data parallelism (training.py:73-80 sorts the latent codes of ONE
    batch): the latent codes of the global batch are all-gathered in row order, rank 0's prior sample and projections are broadcast,
    the regulariser is evaluated replicated and every rank injects its rows of dL/dz -- loss curve and parameters equal the
    single-process run with the same draws (fp64; the weight-gradient sums are re-associated across ranks)."""
    script = tmp_path / "w.py"
    script.write_text(_SWAE_DP_WORKER)
    for world in (1, 2):
        env = dict(os.environ, REPO=REPO, OUT=str(tmp_path), BALER_AMD_FORCE_DEVICE="0", BALER_AMD_DIST_BACKEND="gloo")
        for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "BALER_AMD_FORCE_PG"):
            env.pop(k, None)
        cmd = [sys.executable, str(script)] if world == 1 else \
            [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr", "127.0.0.1",
             "--master-port", str(free_port()), str(script)]
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    l1 = np.load(tmp_path / "w1r0" / "loss_data.npy")
    l2 = np.load(tmp_path / "w2r0" / "loss_data.npy")
    assert l1.shape == (2, 2) and np.all(np.isfinite(l1)) and np.allclose(l2, l1, rtol=1e-9, atol=0)
    p1 = np.load(tmp_path / "swae_params_w1r0.npy")
    p20, p21 = np.load(tmp_path / "swae_params_w2r0.npy"), np.load(tmp_path / "swae_params_w2r1.npy")
    assert np.array_equal(p20, p21)
    assert np.linalg.norm(p20 - p1) / np.linalg.norm(p1) < 1e-9
