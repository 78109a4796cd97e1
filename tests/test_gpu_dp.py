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
    r = _torchrun(2, [str(script)], env, 29541)
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
                      "--no-extras"], env, 29542)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1                                # rank 0 prints ONE JSON line
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and "cpu_baseline" not in d and d["roofline"]["frac"] > 0
