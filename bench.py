#!/usr/bin/env python3
"""bench.py -- rows/sec of the Baler dense-AE hot path on MI355X (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

Workload (BASELINE.json configs[1]): CMS 24-column AE 24->200->100->50->15->50->100->200->24,
1,000,000 synthetic CMS-like rows per GPU resident in HBM (float64, min-max normalised, as
training.train keeps them).  One "step" = one pass of the training hot path over that batch:
forward + sum-of-squares loss + backward over the rank's 1M rows, ONE RCCL sum-all-reduce of the flat
[grads | loss] buffer (N > 1), one fused Adam step.  Weak scaling: rows per GPU fixed, global batch =
N x 1M rows.  `value` = N * rows * K / time (max over ranks, barrier + synchronize on both sides).

Extra keys on the same JSON line: encode / decode rows/s (same resident rows), the strict reference
batching regime (batch_size 512, sequential steps, latency-bound), `roofline` for the dominant kernel
(HIP events on the launch stream, inside the timed region) and `cpu_baseline` (the plain-PyTorch fp64
restatement of the reference loops, timed on this box's host cores; N == 1 only).
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
if REPO not in sys.path:
    sys.path.insert(0, REPO)



def _spawn_ranks_if_needed():
    """`python bench.py --gpus N` (N > 1) outside torchrun: start the N ranks OURSELVES.  The parent never touches
    the GPU (no torch import, no HIP call): it runs `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    bench.py <same args>` as a CHILD process, relays its output (rank 0 prints the one JSON line) and exits with the
    child's return code.  Under torchrun (WORLD_SIZE set) this is a no-op."""
    if "WORLD_SIZE" in os.environ:
        return
    n = 1
    for i, a in enumerate(sys.argv):
        if a == "--gpus" and i + 1 < len(sys.argv):
            n = int(sys.argv[i + 1])
        elif a.startswith("--gpus="):
            n = int(a.split("=", 1)[1])
    if n <= 1:
        return
    import socket
    import subprocess
    port = os.environ.get("MASTER_PORT")
    if not port:
        with socket.socket() as s:      # a free port: two benches on one host must not collide
            s.bind(("127.0.0.1", 0))
            port = str(s.getsockname()[1])
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")   # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}",
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + sys.argv[1:]
    sys.exit(subprocess.run(cmd, env=env).returncode)


if __name__ == "__main__":
    _spawn_ranks_if_needed()

import numpy as np
import torch

# algorithmic (unpadded) work per row: SURVEY.md section 8(d) / BASELINE.md section 4
FLOP_TRAIN_ROW = 357_000
FLOP_ENCODE_ROW = 61_100
BYTES_TRAIN_ROW = 192            # the float64 input row, read once (algorithmic)
BYTES_ENCODE_ROW = 192 + 120     # float64 row in, float64 latent row out
FLOP_C4_ENCODE, FLOP_C4_TRAIN, FLOP_C5_ENCODE = 1_052_500, 5_315_000, 255_400
PEAK_TFLOPS = {"fp32": 157.3, "fp64": 78.6, "bf16": 2500.0}        # dense MFMA peaks, MI355X_MICROARCH.md
PEAK_HBM_GBS = 8000.0
DTYPE_NAME = {"fp32": "f32", "fp64": "f64", "bf16": "bf16"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000, help="rows per GPU")
    ap.add_argument("--mode", default=os.environ.get("BALER_AMD_MODE", "fp32"), choices=["fp32", "fp64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip encode/decode/small-batch/bf16/C4/C5 side measurements")
    ap.add_argument("--cpu-rows", type=int, default=1_000_000, help="rows of the CPU baseline's epoch (bounded to ~20 s)")
    ap.add_argument("--pcie-rows", type=int, default=10_000_000, help="rows of the file of the PCIe-inclusive compress / decompress leg")
    ap.add_argument("--c3", action="store_true", help="also run the BASELINE configs[2] leg at --gpus 1 (one 12.5 M-row shard)")
    ap.add_argument("--c3-rows", type=int, default=0, help="rows per GPU of the configs[2] leg (default 100 M / n_gpus, at most 25 M)")
    return ap.parse_args()


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


def rank_diag(stage, local=None):
    """One stderr line per rank and stage, BEFORE the first collective: which device, how many XCDs / CUs, the IPC mode RCCL
    depends on on this driver (HSA_ENABLE_IPC_MODE_LEGACY=0 = dmabuf), visibility masks and the rendezvous -- so that a failed
    multi-GPU run is diagnosable from its tail.  Counting devices does not initialise the GPU; the device properties are only
    read once the process group is up (local is not None)."""
    env = os.environ
    msg = (f"rank {env.get('RANK', '0')}/{env.get('WORLD_SIZE', '1')} local {env.get('LOCAL_RANK', '0')} pid {os.getpid()} {stage}: "
           f"visible_devices={torch.cuda.device_count()} HSA_ENABLE_IPC_MODE_LEGACY={env.get('HSA_ENABLE_IPC_MODE_LEGACY', '<unset>')} "
           f"HIP_VISIBLE_DEVICES={env.get('HIP_VISIBLE_DEVICES', '<unset>')} ROCR_VISIBLE_DEVICES={env.get('ROCR_VISIBLE_DEVICES', '<unset>')} "
           f"master={env.get('MASTER_ADDR', '<unset>')}:{env.get('MASTER_PORT', '<unset>')} "
           f"backend={env.get('BALER_AMD_DIST_BACKEND', 'nccl(RCCL)')} force_device={env.get('BALER_AMD_FORCE_DEVICE', '<unset>')}")
    if local is not None and torch.cuda.is_available():
        p = torch.cuda.get_device_properties(local)
        cus = p.multi_processor_count
        msg += (f" | device {local}: {p.name} arch={getattr(p, 'gcnArchName', '?')} CUs={cus} XCDs={cus // 32 if cus % 32 == 0 else '?'} "
                f"HBM={p.total_memory / 2**30:.0f}GiB")
    log(msg)


def source_hash():
    """Hash of the kernel sources the running library was built from: a committed PMC summary is only quoted when it
    was taken on the same sources."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(REPO, "baler_amd", "csrc", "*.h*"))):
        h.update(open(f, "rb").read())
    return h.hexdigest()[:16]


def timed(fn, steps, world, dev):
    """barrier + synchronize, K calls, synchronize + barrier; returns max-over-ranks seconds."""
    import torch.distributed as td
    if world > 1:
        td.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    torch.cuda.synchronize()
    if world > 1:
        td.barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        td.all_reduce(t, op=td.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def warm(fn, ms=40.0):
    """Run fn back to back for ~`ms` of GPU time before a measurement: the first ~20-30 ms of a new kernel mix run 10-13 % slower than
    its steady state (measured: bf16 fwd_bwd 0.87 ms for the first 20 launches, 0.77 after), whatever ran before.  The number of calls
    depends on the clock: ONLY for functions without collectives (ranks would run different counts); use a fixed count otherwise."""
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    while (time.perf_counter() - t0) * 1e3 < ms:
        for _ in range(4):
            fn()
        torch.cuda.synchronize()


def event_ms(fn, reps):
    """Duration of fn on the launch stream by HIP events (fn must only enqueue work): after warm(), the MEDIAN of five samples of `reps`
    back-to-back launches each -- single samples of a few launches scatter by +-15 % with the clock state (boost after an idle gap,
    power cap under sustained MFMA load)."""
    warm(fn, 25.0)
    samples = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            fn()
        e1.record()
        torch.cuda.synchronize()
        samples.append(e0.elapsed_time(e1) / reps)
    return sorted(samples)[2]


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(rows):
    """The reference's loops restated in plain PyTorch fp64 (oracle/torch_ref.py, bit-identical to the imported
    reference in the authoring container): training.fit with DataLoader + loss.item() per step at batch 512, the
    same epoch on pre-sliced batches (model time without the loader), and helper.compress's encode loop.  A BOUNDED
    sample of the same synthetic workload (~20 s per leg); the rates are per row."""
    from baler_amd import synth
    from oracle import c_oracle as orc
    from oracle import torch_ref
    full = orc.normalize(synth.cms_rows(min(rows, 1_000_000)))
    model = torch_ref.load_flat(torch_ref.DenseAE(24, 15), orc.formula_params(orc.ae_dims(24, 15), 7))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    # 512-row fp64 GEMMs do not scale to every hardware thread of a big host: give the CPU its best
    # intra-op thread count (short calibration), and report the count actually used.
    hw = os.cpu_count() or 1
    best, cores = None, 1
    for nt in sorted({min(hw, c) for c in (4, 8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        cal = torch_ref.make_loader(full[:10240], 512)
        torch_ref.fit_epoch(model, opt, cal)
        t0 = time.perf_counter()
        torch_ref.fit_epoch(model, opt, cal)
        dtc = time.perf_counter() - t0
        log(f"cpu baseline calibration: {nt} threads -> {10240 / dtc:.0f} rows/s")
        if best is None or dtc < best:
            best, cores = dtc, nt
    torch.set_num_threads(cores)
    n = int(min(len(full), max(20480, 20.0 * 10240 / best)))  # bound the timed epoch to ~20 s
    data = full[:n]
    dl = torch_ref.make_loader(data, 512)
    t0 = time.perf_counter()
    torch_ref.fit_epoch(model, opt, dl)
    t_train = time.perf_counter() - t0
    pre = list(torch.tensor(data, dtype=torch.float64).split(512))      # compute-only: batches sliced beforehand
    t0 = time.perf_counter()
    torch_ref.fit_epoch(model, opt, pre)
    t_pre = time.perf_counter() - t0
    n_enc = n // 2
    t0 = time.perf_counter()
    torch_ref.compress_loop(model, data[:n_enc], 512)
    t_enc = time.perf_counter() - t0
    with torch.no_grad():
        xe = torch.tensor(data[:n_enc], dtype=torch.float64)
        t0 = time.perf_counter()
        for b in xe.split(512):
            model.encode(b)
        t_enc_pre = time.perf_counter() - t0
    return {
        "value": n / t_train, "unit": "rows/s", "cores": cores, "kind": "port",
        "train_compute_only_rows_per_s": n / t_pre,
        "encode_rows_per_s": n_enc / t_enc, "encode_compute_only_rows_per_s": n_enc / t_enc_pre,
        "cpu_model": cpu_model(), "hw_threads": hw, "rows_requested": rows, "rows_timed": n,
        "compare_with": "train_rows_per_s_by_batch['512'] (the same batch_size = 512 regime), not `value` "
                        "(one optimizer step per 1M-row batch)",
        "sample": f"{n} of the {rows} requested rows x 24 cols (bounded to ~20 s per leg), 1 epoch of training.fit at "
                  f"batch_size 512 through DataLoader (fp64, torch {torch.__version__}, {cores} of {hw} threads: the best "
                  f"of 4..64), the same epoch on pre-sliced batches, and the encode loop of helper.compress on {n_enc} rows",
    }


def main():
    a = parse()
    from baler_amd import dist as bdist
    from baler_amd import native, synth
    from baler_amd.modules import models

    rank_diag("start")
    try:
        rank, world, local = bdist.init_from_env()
    except Exception as e:      # RCCL / rendezvous failure: say so and fail; never re-exec a process that may have touched the GPU
        print(f"bench.py: torch.distributed initialisation failed on rank {os.environ.get('RANK', '?')}: {type(e).__name__}: {e}",
              file=sys.stderr, flush=True)
        sys.exit(3)
    rank_diag("process group up", local)
    coll = bdist.collectives_on()      # world > 1, or a one-rank group forced with BALER_AMD_FORCE_PG=1 (RCCL's first run on one GPU)
    if rank == 0 and torch.cuda.is_available():        # (the library version is readable without a communicator)
        try:
            log("RCCL version " + ".".join(str(v) for v in torch.cuda.nccl.version()))
        except Exception as e:
            log(f"RCCL version unavailable: {type(e).__name__}: {e}")
    if coll:
        # the FIRST collective of the run, on its own, so that a hang or an IPC failure is attributable from the log tail
        import torch.distributed as td
        try:
            probe = torch.ones(1, device=torch.device("cuda", local) if torch.cuda.is_available() else "cpu")
            td.all_reduce(probe)
            ok = float(probe.item()) == world
        except Exception as e:
            print(f"bench.py: first all-reduce failed on rank {rank}: {type(e).__name__}: {e}", file=sys.stderr, flush=True)
            sys.exit(3)
        rank_diag(f"first all-reduce {'OK' if ok else 'WRONG SUM'}", local)
        if not ok:
            sys.exit(3)
    if world != a.gpus:
        raise SystemExit(f"bench.py: --gpus {a.gpus} but WORLD_SIZE is {world} (launch with --nproc-per-node {a.gpus}, "
                         f"or run `python bench.py --gpus {a.gpus}` and let it start the ranks)")
    native.require_gpu()
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)

    # ---- resident synthetic rows: generated per rank (counter based), normalised on the device -------
    raw = torch.as_tensor(synth.cms_rows(a.rows, row0=rank * a.rows)).to(dev)
    feats = native.minmax(raw)
    x = native.normalize(raw, feats, torch.float64)
    del raw

    torch.manual_seed(0)
    model = models.AE(24, 15, mode=a.mode).to(dev)
    h = model.handle()
    flat = model.flat
    grads, m, v = torch.zeros_like(flat), torch.zeros_like(flat), torch.zeros_like(flat)
    loss_acc = torch.zeros(1, dtype=torch.float64, device=dev)
    state = {"t": 0}
    ev = []  # (start, end) HIP events around the dominant kernel call, on the launch stream

    def train_step(record=False):
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        h.fwd_bwd(x, grads)
        if record:
            e1.record()
            ev.append((e0, e1))
        if coll:
            bdist.allreduce_sum(grads)
        state["t"] += 1
        h.adam_step(flat, grads, m, v, state["t"], 1e-3, loss_accum=loss_acc)

    # The library's own RCCL communicator (bamd_comm_init: fwd_bwd -> ncclAllReduce -> Adam in ONE host call), built in a helper thread
    # with a deadline: should ncclCommInitRank not return on some node (it never failed on the boxes this was developed on), the run says
    # so on the line and measures every data-parallel loop as three Python calls, as rounds 1-5 did.  `value` never depends on it.
    lib_dp, lib_dp_error = False, None
    if coll and torch.distributed.get_backend() == "nccl" and os.environ.get("BALER_AMD_LIB_COMM", "1") != "0":
        import threading
        att = {}

        def _attach():
            try:
                torch.cuda.set_device(local)      # (the current device is per thread: a new thread starts on device 0)
                bdist.attach_comm(h)
                att["ok"] = True
            except Exception as e:      # noqa: BLE001
                att["err"] = f"{type(e).__name__}: {e}"
        th = threading.Thread(target=_attach, daemon=True)
        th.start()
        th.join(timeout=float(os.environ.get("BALER_AMD_COMM_INIT_TIMEOUT", "90")))
        if th.is_alive():
            state["hard_exit"] = True      # a thread stuck inside RCCL: leave through os._exit after the JSON line
            lib_dp_error = "timeout building the library's RCCL communicator"
        elif "err" in att:
            lib_dp_error = att["err"]
        else:
            lib_dp = True
        if world > 1:      # every rank must take the same path through the data-parallel loops
            okf = torch.tensor([1.0 if lib_dp else 0.0], device=torch.device("cuda", local))
            torch.distributed.all_reduce(okf, op=torch.distributed.ReduceOp.MIN)
            if lib_dp and float(okf.item()) == 0.0:
                lib_dp, lib_dp_error = False, "another rank could not build the library's communicator"
        rank_diag("library communicator " + ("up" if lib_dp else f"NOT available ({lib_dp_error})"), local)
    log("data resident, model ready")
    # clocks and caches to their steady state before the W warm-up steps (forward + backward only: the gradient buffer is rewritten by
    # every step, nothing else is touched); see warm()
    warm(lambda: h.fwd_bwd(x, grads), 60.0)
    for _ in range(a.warmup):
        train_step()
    torch.cuda.synchronize()
    log("warm-up done")
    replicas_identical = None
    if world > 1:
        # every rank must hold bit-identical parameters after the replicated Adam steps: MAX - MIN of two checksums
        import torch.distributed as td
        w = torch.arange(1, flat.numel() + 1, dtype=torch.float64, device=dev)
        chk = torch.stack([flat.double().sum(), (flat.double() * w).sum()])
        hi_, lo_ = chk.clone(), chk.clone()
        td.all_reduce(hi_, op=td.ReduceOp.MAX)
        td.all_reduce(lo_, op=td.ReduceOp.MIN)
        replicas_identical = bool(torch.equal(hi_, lo_))
        if not replicas_identical:
            raise SystemExit(f"bench.py: parameter replicas diverged after {a.warmup} steps: checksum spread {(hi_ - lo_).tolist()}")
    dt = timed(lambda: train_step(True), a.steps, world, dev)
    rows_total = world * a.rows * a.steps
    value = rows_total / dt
    k_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    log(f"train: {value:.4g} rows/s, {1e3 * dt / a.steps:.3f} ms/step, fwd_bwd {k_ms:.3f} ms")
    final_loss = float(grads[-1].item())
    wsum = torch.arange(1, flat.numel() + 1, dtype=torch.float64, device=dev)
    param_checksum = [float(flat.double().sum().item()), float((flat.double() * wsum).sum().item())]
    grad_checksum = [float(grads.double().sum().item()), float((grads.double() * wsum).sum().item())]
    del wsum

    out = {
        "metric": "rows/sec (train) + rows/sec (encode), CMS 24-col AE at 1/2/4/8 GPUs",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NAME[a.mode], "data": "synthetic",
        "config": {
            "workload": f"CMS 24-col AE(24,15) train step (fwd+loss+bwd+Adam), {a.rows} synthetic rows per GPU "
                        f"resident in HBM as float64, one optimizer step per pass (global batch = n_gpus x {a.rows}), "
                        f"{a.mode} MFMA (the 1e-5 parity mode), random-init weights",
            "rows_per_gpu": a.rows, "n_cols": 24, "latent": 15, "parallelism": f"dp{world}",
            "compute_mode": a.mode,
        },
        "train_rows_per_s": value, "last_batch_loss": final_loss,
        "rccl_ranks": bdist.rank_world()[1],
        "dist_backend": (torch.distributed.get_backend() if coll else None),
        "replicas_identical": replicas_identical,
        "param_checksum": param_checksum, "grad_checksum": grad_checksum,   # after the K timed steps (float64 sums: bit-exact run to run)
        "source_hash": source_hash(),
    }
    if coll:
        # the 247 KB [grads | loss] all-reduce alone (latency-bound on xGMI), 200 back-to-back calls
        for _ in range(10):
            bdist.allreduce_sum(grads)
        t_ar = timed(lambda: bdist.allreduce_sum(grads), 200, world, dev)
        out["allreduce_us"] = 1e6 * t_ar / 200
        out["allreduce_bytes"] = grads.numel() * grads.element_size()
        out["allreduce_frac_of_step"] = out["allreduce_us"] * 1e-3 / out["ms_per_step"]
        log(f"all-reduce of {out['allreduce_bytes']} B: {out['allreduce_us']:.1f} us")

    achieved = FLOP_TRAIN_ROW * a.rows / (k_ms * 1e-3) / 1e12
    # HBM traffic per launch comes from rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE in separate
    # runs, gfx950 x2 read correction; counters cannot be read in-process).  The committed summary is quoted only when it
    # was taken on the kernel sources this library was built from; otherwise traffic is null.
    traffic = None
    try:
        with open(os.path.join(REPO, "profiles", "pmc_summary.json")) as f:
            pm = json.load(f)
        if pm.get("rows") == a.rows and a.mode == "fp32" and pm.get("source_hash") == out["source_hash"]:
            traffic = pm["fwd_bwd_hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    out["roofline"] = {
        "bound": "mfma",
        "kernel": ("bamd_fwd_bwd = train_dec_kernel + train_enc_kernel + partial-gradient reduction" if a.mode == "fp32" else
                   "bamd_fwd_bwd, BAMD_MODE_F64 = chain64r_kernel + dw64m_kernel per 262,144-row chunk + one finishing dw64_kernel"),
        "achieved": achieved, "peak": PEAK_TFLOPS[a.mode], "unit": "TFLOP/s",
        "frac": achieved / PEAK_TFLOPS[a.mode], "traffic": traffic,
        "launch_ms": k_ms, "algorithmic_flop_per_row": FLOP_TRAIN_ROW, "rows_per_launch": a.rows,
    }
    if a.mode == "fp32":
        # the pair's instruction multiset replayed with every dependency removed (tools/isa_mix.py, profiles/r6_f32_train_mix_replay.txt):
        # 39.0 + 15.0 us per 64-row iteration of a workgroup, i.e. what THIS multiset can issue; the shipped iteration beside it
        iters = a.rows / 64 / 256
        out["roofline"].update({
            "issued_over_algorithmic_macs": 1.16,
            "us_per_64_row_iteration": 1e3 * (k_ms - 0.015) / iters if iters >= 1 else None,
            "mix_replay_us_per_iteration": 54.0,
            "frac_at_mix_replay": (FLOP_TRAIN_ROW * a.rows / ((iters * 54.0 + 15.0) * 1e-6) / 1e12 / PEAK_TFLOPS["fp32"]) if iters >= 1 else None})

    if not a.no_extras:
        extra_roof = {}
        # encode / decode passes over the same resident rows (no collectives: rows shard naturally)
        z = h.encode(x)
        n_enc = max(3, a.steps // 2)
        warm(lambda: h.encode(x))       # (as for the bf16 pair below: 15 launches of 0.5 ms sit inside the clock ramp without it -- one run read 1.29 G rows/s)
        t_enc = timed(lambda: h.encode(x), n_enc, world, dev)
        out["encode_rows_per_s"] = world * a.rows * n_enc / t_enc
        warm(lambda: h.decode(z))
        t_dec = timed(lambda: h.decode(z), n_enc, world, dev)
        out["decode_rows_per_s"] = world * a.rows * n_enc / t_dec
        out["encode_tflops"] = FLOP_ENCODE_ROW * a.rows * n_enc / t_enc / 1e12
        ms = event_ms(lambda: h.encode(x), 5)
        extra_roof["encode_" + DTYPE_NAME[a.mode]] = {
            "bound": "mfma", "kernel": "bamd_encode (infer2_kernel)", "launch_ms": ms, "unit": "TFLOP/s",
            "achieved": FLOP_ENCODE_ROW * a.rows / ms / 1e9, "peak": PEAK_TFLOPS[a.mode],
            "frac": FLOP_ENCODE_ROW * a.rows / ms / 1e9 / PEAK_TFLOPS[a.mode],
            "hbm_gbs_algorithmic": BYTES_ENCODE_ROW * a.rows / ms / 1e6, "hbm_frac": BYTES_ENCODE_ROW * a.rows / ms / 1e6 / PEAK_HBM_GBS}
        if a.mode == "fp32":
            # ---- bf16 mode on the same rows and weights: a THROUGHPUT mode with its own 2e-2 bar (tests), never `value` ----
            hb = native.Handle(model.dims, "bf16")
            hb.load_params(flat)
            zb = hb.encode(x)
            out["bf16_encode_rel_err_vs_fp32"] = float((zb.double() - z.double()).norm() / z.double().norm())
            warm(lambda: hb.encode(x))
            t_b = timed(lambda: hb.encode(x), n_enc, world, dev)
            out["bf16_encode_rows_per_s"] = world * a.rows * n_enc / t_b
            warm(lambda: hb.decode(z))
            t_b = timed(lambda: hb.decode(z), n_enc, world, dev)
            out["bf16_decode_rows_per_s"] = world * a.rows * n_enc / t_b
            ms = event_ms(lambda: hb.encode(x), 5)
            extra_roof["encode_bf16"] = {
                "bound": "hbm", "kernel": "bamd_encode (bf16_infer_kernel)", "launch_ms": ms, "unit": "GB/s",
                "achieved": BYTES_ENCODE_ROW * a.rows / ms / 1e6, "peak": PEAK_HBM_GBS,
                "frac": BYTES_ENCODE_ROW * a.rows / ms / 1e6 / PEAK_HBM_GBS,
                "mfma_tflops": FLOP_ENCODE_ROW * a.rows / ms / 1e9, "mfma_frac": FLOP_ENCODE_ROW * a.rows / ms / 1e9 / PEAK_TFLOPS["bf16"],
                # what actually bounds this kernel: the SIMD's issue port, not HBM.  Its layers issue 1.34 x the
                # algorithmic MACs (K padded to 32) beside 3.4 VALU instructions per MFMA (LeakyReLU = multiply + maximum + half a conversion per
                # activation value: gfx950 has no packed-bf16 arithmetic); tools/probe/mfma_shape_probe.hip issues 1,033 - 1,219 TFLOP/s at
                # 3.75 - 2.5 VALU per MFMA (profiles/r3_mfma_shape_probe.txt), i.e. ~1,085 at this kernel's ratio
                "issued_tflops": 1.34 * FLOP_ENCODE_ROW * a.rows / ms / 1e9, "issue_ceiling_tflops": 1085.0,
                "frac_of_issue_ceiling": 1.34 * FLOP_ENCODE_ROW * a.rows / ms / 1e9 / 1085.0}
            # bf16 MFMA TRAINING (BASELINE configs[1] names bf16): fp32 master weights + fp32 Adam, bf16 kernels
            fb = flat.clone()
            gb, mb, vb = torch.zeros_like(fb), torch.zeros_like(fb), torch.zeros_like(fb)
            hb.load_params(fb)
            tb = {"t": 0}

            def bf16_step():
                hb.fwd_bwd(x, gb)
                if world > 1:
                    bdist.allreduce_sum(gb)
                tb["t"] += 1
                hb.adam_step(fb, gb, mb, vb, tb["t"], 1e-3)
            for _ in range(50):      # ~40 ms to the steady state; a fixed count: the step holds a collective when world > 1
                bf16_step()
            t_bt = timed(bf16_step, max(5, a.steps // 2), world, dev)
            out["bf16_train_rows_per_s"] = world * a.rows * max(5, a.steps // 2) / t_bt
            ms = event_ms(lambda: hb.fwd_bwd(x, gb), 20)
            extra_roof["train_bf16"] = {
                "bound": "mfma", "kernel": "bamd_fwd_bwd, BAMD_MODE_BF16 = bf16_train_kernel<PART 0> + <PART 1> + reduce_tiles_k",
                "launch_ms": ms, "unit": "TFLOP/s", "achieved": FLOP_TRAIN_ROW * a.rows / ms / 1e9, "peak": PEAK_TFLOPS["bf16"],
                "frac": FLOP_TRAIN_ROW * a.rows / ms / 1e9 / PEAK_TFLOPS["bf16"],
                "hbm_gbs_algorithmic": BYTES_TRAIN_ROW * a.rows / ms / 1e6, "hbm_frac": BYTES_TRAIN_ROW * a.rows / ms / 1e6 / PEAK_HBM_GBS,
                "last_batch_loss": float(gb[-1].item()),
                # issued MACs are 1.38 x the algorithmic ones (K padded to 32); the pair's main-loop instruction multiset replayed with every
                # dependency removed (tools/isa_mix.py, profiles/r6_bf16_train_mix_replay.txt) takes 12.1 - 12.7 us per 64-row iteration of
                # both launches: the schedulable ceiling of THIS multiset
                "issued_frac": 1.38 * FLOP_TRAIN_ROW * a.rows / ms / 1e9 / PEAK_TFLOPS["bf16"],
                "us_per_64_row_iteration": 1e3 * (ms - 0.017) / (a.rows / 64 / 256), "mix_replay_us_per_iteration": [12.1, 12.7]}
            if world == 1:      # the bf16 handle's optimizer steps by batch size (<= 3072 rows: the fp32 small-batch kernels)
                bb = {}
                for bs in (512, 8192, 32768):
                    if bs * 2 > a.rows:
                        continue
                    nb = max(2, min(200, a.rows // bs))

                    def bpass():
                        for i in range(nb):
                            tb["t"] += 1
                            hb.train_step(x[i * bs:(i + 1) * bs], fb, mb, vb, tb["t"], 1e-3)
                    bpass()
                    t_ = timed(bpass, 1, world, dev)
                    bb[str(bs)] = {"rows_per_s": bs * nb / t_, "us_per_step": 1e6 * t_ / nb}
                out["bf16_train_rows_per_s_by_batch"] = bb
                # (the two round-5 rewrites that rode this line in round 5 -- 0.833 / 0.984 ms against the pair's 0.744 -- were removed
                # in round 6: profiles/r5_bench.json, profiles/r5_bf16_regchain_kernel_stats.csv, profiles/r5_bf16_quad_kernel_stats.csv)
            hb.close()
        # ---- the reference's batching regime and the curve up to the benchmarked batch: sequential optimizer steps ----
        by_batch = {}
        for bs in (512, 4096, 32768, 262144):
            nb = max(2, min(400, a.rows // bs))
            if bs * 2 > a.rows:
                continue

            def batch_pass():
                if lib_dp:          # what training.fit issues under RCCL: ONE bamd_train_epoch_dp per epoch and rank
                    state["t"] += h.train_epoch_dp(x[:nb * bs], [bs] * nb, flat, m, v, state["t"] + 1, 1e-3, loss_accum=loss_acc, grads=grads)
                    return
                for i in range(nb):
                    state["t"] += 1
                    xb = x[i * bs:(i + 1) * bs]
                    if world == 1:      # what training.fit issues: one bamd_train_step per batch
                        h.train_step(xb, flat, m, v, state["t"], 1e-3, loss_accum=loss_acc)
                        continue
                    h.fwd_bwd(xb, grads)
                    bdist.allreduce_sum(grads)
                    h.adam_step(flat, grads, m, v, state["t"], 1e-3, loss_accum=loss_acc)
            for _ in range(4 if bs <= 4096 else 1):      # to the steady state (fixed count: collectives inside when world > 1)
                batch_pass()
            tb_ = timed(batch_pass, 1, world, dev)
            by_batch[str(bs)] = {"rows_per_s": world * bs * nb / tb_, "us_per_step": 1e6 * tb_ / nb, "steps_timed": nb,
                                 "tflops": FLOP_TRAIN_ROW * world * bs * nb / tb_ / 1e12,
                                 "frac_of_mfma_peak": FLOP_TRAIN_ROW * bs * nb / tb_ / 1e12 / PEAK_TFLOPS[a.mode]}
        out["train_rows_per_s_by_batch"] = by_batch
        if coll:
            out["dp_steps_run_by"] = "library (bamd_train_epoch_dp)" if lib_dp else "python (fwd_bwd, torch all-reduce, adam_step)"
        if world > 1:
            # the numbers above take bs rows PER GPU per step (dp_batch = "per_gpu": global batch = N x bs); the reference's
            # config means the GLOBAL batch (dist.batch_policy's default): every rank computes bs / N rows of each step
            out["train_rows_per_s_by_batch_policy"] = "per_gpu"
            by_global = {}
            for bs in (512, 4096, 32768):
                per = bs // world
                if per < 16 or per * world != bs:
                    continue
                nb = max(2, min(400, a.rows // per))

                def gpass():
                    if lib_dp:
                        state["t"] += h.train_epoch_dp(x[:nb * per], [per] * nb, flat, m, v, state["t"] + 1, 1e-3, loss_accum=loss_acc, grads=grads)
                        return
                    for i in range(nb):
                        state["t"] += 1
                        h.fwd_bwd(x[i * per:(i + 1) * per], grads)
                        bdist.allreduce_sum(grads)
                        h.adam_step(flat, grads, m, v, state["t"], 1e-3, loss_accum=loss_acc)
                gpass()
                tg_ = timed(gpass, 1, world, dev)
                by_global[str(bs)] = {"rows_per_s": bs * nb / tg_, "us_per_step": 1e6 * tg_ / nb, "rows_per_gpu_per_step": per,
                                      "steps_timed": nb}
            out["train_rows_per_s_by_global_batch"] = by_global
        if coll and torch.distributed.get_backend() == "nccl" and os.environ.get("BALER_AMD_LIB_COMM", "1") != "0":
            # ---- what a data-parallel optimiser step costs, as the three-call Python sequence (bamd_fwd_bwd -> torch all-reduce ->
            # bamd_adam_step) and with the communicator INSIDE the library (bamd_train_epoch_dp: the same three stages per batch, one
            # host call per epoch), at 64 and 512 rows per rank.  gpu_us = wall time per step with the stream drained; host_us = the
            # host thread's time per step until its last enqueue returns.
            dp = {}
            if not lib_dp:
                dp["error"] = lib_dp_error
            else:
                g2 = torch.zeros_like(flat)
                for _ in range(10):
                    h.allreduce_sum(g2)
                dp["lib_allreduce_us"] = 1e6 * timed(lambda: h.allreduce_sum(g2), 200, world, dev) / 200
                for per in (64, 512):
                    nb = max(2, min(400, a.rows // per))

                    def py_steps():
                        for i in range(nb):
                            state["t"] += 1
                            h.fwd_bwd(x[i * per:(i + 1) * per], grads)
                            bdist.allreduce_sum(grads)
                            h.adam_step(flat, grads, m, v, state["t"], 1e-3, loss_accum=loss_acc)

                    def lib_steps():
                        state["t"] += h.train_epoch_dp(x[:nb * per], [per] * nb, flat, m, v, state["t"] + 1, 1e-3, loss_accum=loss_acc, grads=grads)
                    rec = {"rows_per_rank": per, "steps_timed": nb}
                    for tag, fn in (("python_3_calls", py_steps), ("library_1_call", lib_steps)):
                        fn(); fn()
                        torch.cuda.synchronize()
                        if world > 1:
                            torch.distributed.barrier()
                        t0 = time.perf_counter()
                        fn()
                        t_host = time.perf_counter() - t0
                        torch.cuda.synchronize()
                        t_all = time.perf_counter() - t0
                        rec[tag] = {"gpu_us": 1e6 * t_all / nb, "host_us": 1e6 * t_host / nb}
                    dp[str(per)] = rec
            out["dp_step"] = dp
            log("dp step: " + json.dumps(dp))
        if world == 1 and not coll and a.rows >= 2 * 512:
            # ---- the reference's epoch (training.fit, training.py:64-97: sequential batches of 512 rows) as ONE host call: bamd_train_epoch,
            # the batch loop inside the library.  host_us_per_step = the host thread's time inside the call (the enqueue of two
            # launches per step); us_per_step = epoch wall time / steps with the stream drained (GPU-bound when larger than the host's)
            nb = a.rows // 512

            def epoch():
                state["t"] += h.train_epoch(x[:nb * 512], 512, flat, m, v, state["t"] + 1, 1e-3, loss_accum=loss_acc)
            epoch()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            epoch()
            t_host = time.perf_counter() - t0
            torch.cuda.synchronize()
            t_all = time.perf_counter() - t0
            out["train_bs512_epoch_call"] = {"steps": nb, "host_calls": 1, "us_per_step": 1e6 * t_all / nb, "host_us_per_step": 1e6 * t_host / nb,
                                             "rows_per_s": nb * 512 / t_all}
        if "512" in by_batch:
            out["train_bs512_rows_per_s"] = by_batch["512"]["rows_per_s"]
            out["train_bs512_us_per_step"] = by_batch["512"]["us_per_step"]
            extra_roof["train_bs512_" + DTYPE_NAME[a.mode]] = {
                "bound": "latency", "kernel": "bamd_train_step at 512 rows (lat4_chain_kernel + lat2_dw_kernel<adam>)",
                "launch_us": by_batch["512"]["us_per_step"] / 1.0, "unit": "TFLOP/s",
                "achieved": by_batch["512"]["tflops"] / world, "peak": PEAK_TFLOPS[a.mode],
                "frac": by_batch["512"]["frac_of_mfma_peak"]}
        if world == 1 and a.mode == "fp32":
            # ---- fp64 mode (the reference's own dtype, models.py:128-136): fused small-batch step, layer-wise inference / large batches ----
            h64 = native.Handle(model.dims, "fp64")
            f64 = flat.double().clone()
            h64.load_params(f64)
            m64, v64 = torch.zeros_like(f64), torch.zeros_like(f64)
            g64 = torch.zeros_like(f64)
            n64 = min(a.rows, 262144)
            ms = event_ms(lambda: h64.encode(x[:n64]), 3)
            extra_roof["encode_f64"] = {"bound": "mfma", "kernel": "bamd_encode, BAMD_MODE_F64 (infer64_kernel: register chain on v_mfma_f64_16x16x4_f64)", "rows": n64,
                                        "launch_ms": ms, "unit": "TFLOP/s", "achieved": FLOP_ENCODE_ROW * n64 / ms / 1e9, "peak": PEAK_TFLOPS["fp64"],
                                        "frac": FLOP_ENCODE_ROW * n64 / ms / 1e9 / PEAK_TFLOPS["fp64"], "rows_per_s": n64 / ms * 1e3}
            # the headline workload (all of the rank's rows in one step) in the reference's own dtype: chunk after chunk of 262,144 rows
            # over one image buffer (fused64.hip, round 4); and the 65,536-row point of rounds 2-3 beside it
            for key, n64t in (("train_f64", a.rows), ("train_f64_64k", min(a.rows, 65536))):
                ms = event_ms(lambda: h64.fwd_bwd(x[:n64t], g64), 3)
                extra_roof[key] = {"bound": "mfma", "kernel": "bamd_fwd_bwd, BAMD_MODE_F64 (chain64r_kernel (one wave per 16-row block, activations in registers; chain64_kernel below 1,024 blocks) + dw64m_kernel per 262,144-row chunk + one finishing dw64_kernel: the fused pair, weight-gradient tiles in 16-tile blocks oriented per layer from 16,384 rows on, 2 x 4 blocks below)", "rows": n64t,
                                   "launch_ms": ms, "unit": "TFLOP/s", "achieved": FLOP_TRAIN_ROW * n64t / ms / 1e9, "peak": PEAK_TFLOPS["fp64"],
                                   "frac": FLOP_TRAIN_ROW * n64t / ms / 1e9 / PEAK_TFLOPS["fp64"], "rows_per_s": n64t / ms * 1e3}
            t64 = {"t": 0}

            def steps64():
                for i in range(100):
                    t64["t"] += 1
                    h64.train_step(x[i * 512:(i + 1) * 512], f64, m64, v64, t64["t"], 1e-3)
            steps64()
            us = 1e6 * timed(steps64, 1, world, dev) / 100
            extra_roof["train_bs512_f64"] = {"bound": "latency", "kernel": "bamd_train_step at 512 rows, BAMD_MODE_F64 (chain64_kernel + dw64_kernel<adam>)",
                                             "launch_us": us, "unit": "TFLOP/s", "achieved": FLOP_TRAIN_ROW * 512 / us / 1e6, "peak": PEAK_TFLOPS["fp64"],
                                             "frac": FLOP_TRAIN_ROW * 512 / us / 1e6 / PEAK_TFLOPS["fp64"], "rows_per_s": 512 / us * 1e6}
            h64.close()
        out["roofline_extra"] = extra_roof
        log(f"encode {out['encode_rows_per_s']:.4g} rows/s, decode {out['decode_rows_per_s']:.4g} rows/s, "
            + ", ".join(f"bs{k} {v_['us_per_step']:.1f} us" for k, v_ in by_batch.items())
            + (f", bf16 train {out['bf16_train_rows_per_s']:.4g} rows/s" if "bf16_train_rows_per_s" in out else ""))
        if rank == 0 and world == 1 and a.mode == "fp32":
            out["other_configs"] = other_configs(dev)
            try:
                out["pcie"] = pcie_leg(dev, h, a.pcie_rows)
                out["pcie_compress_rows_per_s"] = out["pcie"]["compress_rows_per_s"]
                out["pcie_decompress_rows_per_s"] = out["pcie"]["decompress_rows_per_s"]
            except OSError as e:       # no scratch space for the file: say so, the line stays valid
                out["pcie"] = {"error": str(e)}

    if world > 1 or a.c3:
        out["c3"] = c3_leg(a, h, flat, m, v, state, world, dev, rank, loss_acc)
        out["c3_train_rows_per_s"] = out["c3"]["train_rows_per_s"]

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_rows)

    if rank == 0:
        print(json.dumps(out), flush=True)
    if state.get("hard_exit"):
        sys.stdout.flush(); sys.stderr.flush()
        os._exit(0)
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


def c3_leg(a, h, flat, m, v, state, world, dev, rank, loss_acc):
    """BASELINE configs[2]: the 100 M-row table, data parallel: every rank holds 100 M / N rows resident (12.5 M at N = 8;
    capped at 25 M rows = 4.8 GB so that N = 2 stays bounded -- `exact_config` says whether the cap applied), one optimizer
    step per pass over the shards with the RCCL all-reduce between bamd_fwd_bwd and bamd_adam_step.  Rows come from the
    same counter-based generator, evaluated on the device."""
    from baler_amd import dist as bdist
    from baler_amd import native, synth
    want = 100_000_000 // world if world > 1 else 12_500_000
    rows = a.c3_rows or min(want, 25_000_000)
    raw = synth.cms_rows_torch(rows, row0=rank * rows, device=dev)
    mm = native.col_minmax(raw)      # [min ; max] of this shard; the table's extrema by one MIN and one MAX all-reduce
    bdist.allreduce_minmax(mm)
    feats = torch.stack([mm[0], mm[1] - mm[0]])
    x3 = native.normalize(raw, feats, torch.float64)
    del raw
    g3 = torch.zeros_like(flat)
    ev = []

    def step(record=False):
        if record:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
        h.fwd_bwd(x3, g3)
        if record:
            e1.record()
            ev.append((e0, e1))
        if world > 1:
            bdist.allreduce_sum(g3)
        state["t"] += 1
        h.adam_step(flat, g3, m, v, state["t"], 1e-3, loss_accum=loss_acc)
    step()
    k = 3
    dt = timed(lambda: step(True), k, world, dev)
    k_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    ach = FLOP_TRAIN_ROW * rows / (k_ms * 1e-3) / 1e12
    res = {"rows_per_gpu": rows, "global_rows": rows * world, "exact_config": bool(world == 8 and rows == 12_500_000),
           "steps": k, "ms_per_step": 1e3 * dt / k, "train_rows_per_s": world * rows * k / dt,
           "last_batch_loss": float(g3[-1].item()),
           "roofline": {"bound": "mfma", "kernel": "bamd_fwd_bwd", "launch_ms": k_ms, "achieved": ach, "peak": PEAK_TFLOPS[a.mode],
                        "unit": "TFLOP/s", "frac": ach / PEAK_TFLOPS[a.mode], "traffic": None}}
    log(f"configs[2] leg: {rows} rows per GPU x {world}: {res['train_rows_per_s']:.4g} rows/s, {res['ms_per_step']:.2f} ms/step")
    del x3, g3
    return res


def pcie_leg(dev, h, n):
    """PCIe-INCLUSIVE rates of the compress / decompress data path (never `value`: that is measured with the rows resident):
    an n-row .npz on local disk (page cache warm) -> memory map -> pinned double-buffered H2D -> column min/max ->
    bamd_encode per 4 M-row block with the download of block k overlapping the encode of block k + 1 -> host array; and the
    reverse: latent rows up, bamd_decode (+ un-normalise), decoded table down.  Host clock, second pass (staging pinned)."""
    import shutil
    import tempfile
    from baler_amd import hostio, native, synth
    tmp = tempfile.mkdtemp(prefix="baler_pcie_")
    try:
        path = os.path.join(tmp, "data.npz")
        raw = synth.cms_rows_torch(n, device=dev)
        host = hostio.download_rows(raw)
        del raw
        np.savez(path, data=host, names=synth.CMS_NAMES)
        del host
        res = {}
        B = 1 << 22
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            src = hostio.open_npz_array(path, "data")
            x = hostio.upload_rows(src, None, dev)
            torch.cuda.synchronize(); t1 = time.perf_counter()
            feats = native.minmax(x)
            zdev = torch.empty((n, 15), dtype=torch.float64, device=dev)
            ready = []
            for s0 in range(0, n, B):
                e0 = min(s0 + B, n)
                h.encode(x[s0:e0], features=feats, out=zdev[s0:e0])
                ev = torch.cuda.Event(); ev.record(); ready.append((e0, ev))
            z = hostio.download_rows(zdev, ready=ready)
            torch.cuda.synchronize(); t2 = time.perf_counter()
            del x, zdev
            zd = hostio.upload_rows(z, None, dev)
            dec = torch.empty((n, 24), dtype=torch.float64, device=dev)
            ready = []
            for s0 in range(0, n, B):
                e0 = min(s0 + B, n)
                h.decode(zd[s0:e0], features=feats, out=dec[s0:e0])
                ev = torch.cuda.Event(); ev.record(); ready.append((e0, ev))
            back = hostio.download_rows(dec, ready=ready)
            torch.cuda.synchronize(); t3 = time.perf_counter()
            res = {"rows": n, "file_gb": n * 192 / 1e9, "compress_rows_per_s": n / (t2 - t0), "decompress_rows_per_s": n / (t3 - t2),
                   "h2d_gbs": n * 192 / 1e9 / (t1 - t0), "encode_plus_d2h_gbs_of_latents": n * 120 / 1e9 / (t2 - t1),
                   "decompress_gbs_moved": n * (120 + 192) / 1e9 / (t3 - t2), "finite": bool(np.isfinite(back[:1000]).all())}
            del zd, dec, z, back      # (unmapping a 1.9 GB host array takes ~50 ms: outside the timed regions)
        log(f"PCIe-inclusive: compress {res['compress_rows_per_s']:.4g} rows/s, decompress {res['decompress_rows_per_s']:.4g} rows/s")
        return res
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def other_configs(dev):
    """BASELINE.json configs[3] (CFD 2-D field -> CFD_dense_AE(2500, 25)) and configs[4] (512-column table, encode only)
    on one GPU, fp32 MFMA, so the driver's line carries them too (they are parity-test cases, not `value`)."""
    from baler_amd import synth
    from baler_amd.modules import models
    res = {}
    n = 32768
    xc = torch.as_tensor(synth.cfd_field(n).reshape(n, 2500).astype(np.float32)).to(dev)
    torch.manual_seed(0)
    mc = models.CFD_dense_AE(2500, 25, mode="fp32").to(dev)
    hc = mc.handle()
    gc = torch.zeros_like(mc.flat)
    ms_e = event_ms(lambda: hc.encode(xc), 3)
    zc = hc.encode(xc)
    ms_d = event_ms(lambda: hc.decode(zc), 3)
    ms_t = event_ms(lambda: hc.fwd_bwd(xc, gc), 3)
    res["c4_cfd_dense_2500_25"] = {
        "frames": n, "encode_rows_per_s": n / ms_e * 1e3, "encode_frac_of_mfma_peak": FLOP_C4_ENCODE * n / ms_e / 1e9 / PEAK_TFLOPS["fp32"],
        "decode_rows_per_s": n / ms_d * 1e3, "decode_frac_of_mfma_peak": FLOP_C4_ENCODE * n / ms_d / 1e9 / PEAK_TFLOPS["fp32"],
        "train_fwd_bwd_rows_per_s": n / ms_t * 1e3, "train_frac_of_mfma_peak": FLOP_C4_TRAIN * n / ms_t / 1e9 / PEAK_TFLOPS["fp32"]}
    # the reference's own batch sizes for this model (CFD_project_still: batch_size = 60, CFD_project_animation: 6000): optimiser steps
    pc, mo_, vo_ = mc.flat.clone(), torch.zeros_like(mc.flat), torch.zeros_like(mc.flat)
    stc = {"t": 0}
    by = {}
    for bs in (60, 6000):
        def csteps():
            for i in range(20):
                stc["t"] += 1
                hc.train_step(xc[(i % 5) * bs:(i % 5 + 1) * bs], pc, mo_, vo_, stc["t"], 1e-3)
        csteps()
        by[str(bs)] = {"us_per_step": event_ms(csteps, 3) * 1e3 / 20}
        by[str(bs)]["rows_per_s"] = bs / by[str(bs)]["us_per_step"] * 1e6
    res["c4_cfd_dense_2500_25"]["train_step_by_reference_batch_size"] = by
    hc.load_params(mc.flat)
    hc.close()
    # ... and in the reference's own arithmetic (float64: the layer-wise kernels, one 16 x 16 tile per workgroup at these sizes)
    from baler_amd import native as _native
    h64 = _native.Handle([2500, 200, 100, 50, 25, 50, 100, 200, 2500], "fp64")
    p64 = mc.flat.double()
    h64.load_params(p64)
    m64, v64 = torch.zeros_like(p64), torch.zeros_like(p64)
    x64 = xc[:300].double()

    def c64steps():
        for i in range(10):
            stc["t"] += 1
            h64.train_step(x64[(i % 5) * 60:(i % 5 + 1) * 60], p64, m64, v64, stc["t"], 1e-3)
    c64steps()
    res["c4_cfd_dense_2500_25"]["train_step_by_reference_batch_size"]["60_float64"] = {"us_per_step": event_ms(c64steps, 3) * 1e3 / 10}
    h64.close()
    del p64, m64, v64, x64
    del gc, pc, mo_, vo_
    # the same model in the bf16 mode: en1 / de4 on the bf16 MFMA, HBM-bound (10 KB of float32 per frame)
    from baler_amd import native
    hb = native.Handle([2500, 200, 100, 50, 25, 50, 100, 200, 2500], "bf16")
    hb.load_params(mc.flat)
    zb = hb.encode(xc, out_dtype=torch.float32)
    ms_be = event_ms(lambda: hb.encode(xc, out_dtype=torch.float32), 3)
    ms_bd = event_ms(lambda: hb.decode(zb), 3)
    gb = torch.zeros_like(mc.flat)
    ms_bt = event_ms(lambda: hb.fwd_bwd(xc, gb), 3)      # en1 / de4 / de4's input-gradient product and the two wide weight gradients on the bf16 MFMA
    res["c4_cfd_dense_2500_25"].update({
        "bf16_train_fwd_bwd_rows_per_s": n / ms_bt * 1e3, "bf16_train_vs_fp32": ms_t / ms_bt,
        "bf16_encode_rows_per_s": n / ms_be * 1e3, "bf16_encode_frac_of_hbm": 10100 * n / ms_be / 1e6 / PEAK_HBM_GBS,
        "bf16_decode_rows_per_s": n / ms_bd * 1e3, "bf16_decode_frac_of_hbm": 10100 * n / ms_bd / 1e6 / PEAK_HBM_GBS,
        # what a kernel that does nothing but store row-strided reaches on this part: 4.1 TB/s (tools/probe/hbm_write_probe.hip,
        # profiles/r6_hbm_write_probe.txt; contiguous 1-KiB bursts 4.5 - 5.5); the decode writes 10,000 B per frame
        "bf16_decode_frac_of_store_only_kernel": 10000 * n / ms_bd / 1e6 / 4100.0,
        "bf16_encode_rel_err_vs_fp32": float(torch.linalg.norm(zb.double() - zc.double()) / torch.linalg.norm(zc.double()))})
    hb.close()
    del xc
    # exafel1 / exafel2: 25 x 25 blocks at compression ratio 100 -> CFD_dense_AE(625, 7) (exafel1_config.py:14-15,33)
    n = 131072
    xe = torch.rand((n, 625), dtype=torch.float32, device=dev)
    me = models.CFD_dense_AE(625, 7, mode="fp32").to(dev)
    he = me.handle()
    ge = torch.zeros_like(me.flat)
    ze = he.encode(xe)
    ms_e, ms_d, ms_t = event_ms(lambda: he.encode(xe), 3), event_ms(lambda: he.decode(ze), 3), event_ms(lambda: he.fwd_bwd(xe, ge), 3)
    res["exafel_625_7"] = {
        "blocks": n, "path": he.path, "encode_rows_per_s": n / ms_e * 1e3, "encode_frac_of_mfma_peak": 300_700 * n / ms_e / 1e9 / PEAK_TFLOPS["fp32"],
        "decode_rows_per_s": n / ms_d * 1e3, "decode_frac_of_mfma_peak": 300_700 * n / ms_d / 1e9 / PEAK_TFLOPS["fp32"],
        "train_fwd_bwd_rows_per_s": n / ms_t * 1e3, "train_frac_of_mfma_peak": 1_554_200 * n / ms_t / 1e9 / PEAK_TFLOPS["fp32"]}
    heb = native.Handle([625, 200, 100, 50, 7, 50, 100, 200, 625], "bf16")
    heb.load_params(me.flat)
    ms_bt = event_ms(lambda: heb.fwd_bwd(xe, ge), 3)
    res["exafel_625_7"].update({"bf16_train_fwd_bwd_rows_per_s": n / ms_bt * 1e3, "bf16_train_vs_fp32": ms_t / ms_bt})
    heb.close()
    he.close()
    del xe, ge, ze
    n = 262144
    xw = torch.as_tensor(synth.wide_rows(n, 512).astype(np.float32)).to(dev)
    mw = models.CFD_dense_AE(512, 6, mode="fp32").to(dev)
    hw_ = mw.handle()
    ms = event_ms(lambda: hw_.encode(xw), 3)
    res["c5_encode_512col"] = {"rows": n, "encode_rows_per_s": n / ms * 1e3,
                               "encode_frac_of_mfma_peak": FLOP_C5_ENCODE * n / ms / 1e9 / PEAK_TFLOPS["fp32"],
                               "input_stream_gbs": 2048 * n / ms / 1e6}
    hwb = native.Handle([512, 200, 100, 50, 6, 50, 100, 200, 512], "bf16")
    hwb.load_params(mw.flat)
    ms_b = event_ms(lambda: hwb.encode(xw, out_dtype=torch.float32), 3)
    res["c5_encode_512col"].update({"bf16_encode_rows_per_s": n / ms_b * 1e3, "bf16_encode_frac_of_hbm": (2048 + 24) * n / ms_b / 1e6 / PEAK_HBM_GBS})
    hwb.close()
    hw_.close()
    del xw
    # narrow tables other than the 24-column one (models.py:122-139 builds AE(n_features, z_dim) for any width): class instantiations
    # with run-time widths up to 63 columns / a latent of 31; 64..79 columns fused for inference and small batches; beyond: layer-wise
    n = 1_000_000
    res["narrow_tables"] = {}
    for F, Z in ((30, 8), (47, 31), (63, 31), (64, 16), (80, 16)):
        xn = torch.rand((n, F), dtype=torch.float64, device=dev)
        mn = models.AE(F, Z, mode="fp32").to(dev)
        hn = mn.handle()
        gn = torch.zeros_like(mn.flat)
        mo, vo = torch.zeros_like(mn.flat), torch.zeros_like(mn.flat)
        step = {"t": 0}

        def steps512():
            for i in range(50):
                step["t"] += 1
                hn.train_step(xn[i * 512:(i + 1) * 512], mn.flat, mo, vo, step["t"], 1e-3)
        ms_e, ms_t, ms_s = event_ms(lambda: hn.encode(xn), 3), event_ms(lambda: hn.fwd_bwd(xn, gn), 2), event_ms(steps512, 1) / 50
        res["narrow_tables"][f"ae_{F}_{Z}"] = {"path": hn.path, "encode_rows_per_s": n / ms_e * 1e3, "train_fwd_bwd_rows_per_s": n / ms_t * 1e3,
                                               "train_bs512_us_per_step": ms_s * 1e3}
        hn.close()
        # the same table in the reference's own dtype at the reference's batch size (fp64 class kernels up to 63 columns; 64 .. 127 columns:
        # the 4-row chain, round 6 -- layer-wise before: 134 us)
        m64 = models.AE(F, Z, mode="fp64").to(dev)
        h64n = m64.handle()
        mo64, vo64 = torch.zeros_like(m64.flat), torch.zeros_like(m64.flat)
        st64 = {"t": 0}

        def steps512_64():
            for i in range(50):
                st64["t"] += 1
                h64n.train_step(xn[i * 512:(i + 1) * 512], m64.flat, mo64, vo64, st64["t"], 1e-3)
        res["narrow_tables"][f"ae_{F}_{Z}"]["fp64_train_bs512_us_per_step"] = event_ms(steps512_64, 1) / 50 * 1e3
        h64n.close()
        del xn, gn, mo, vo, mo64, vo64
    # wide models other than the compiled-in shapes (models.py:192-209 builds CFD_dense_AE(n_features, z_dim) for any flattened field):
    # the run-time-width class ImplWide<4096, ZC, true>; on (625, 7) also forced (BALER_AMD_WIDE_CLASS=force at bamd_create) next to
    # the exact instantiation
    res["wide_class"] = {}
    for (F, Z), rows, forced in (((900, 9), 131072, False), ((1024, 11), 131072, False), ((4096, 41), 32768, False), ((128, 13), 524288, False),
                                 ((625, 7), 131072, False), ((625, 7), 131072, True), ((2500, 25), 32768, False), ((2500, 25), 32768, True)):
        if forced:
            os.environ["BALER_AMD_WIDE_CLASS"] = "force"
        try:
            mw = models.CFD_dense_AE(F, Z, mode="fp32").to(dev)
            hw = mw.handle()
        finally:
            os.environ.pop("BALER_AMD_WIDE_CLASS", None)
        xw = torch.rand((rows, F), dtype=torch.float32, device=dev)
        zw = hw.encode(xw)
        gw = torch.zeros_like(mw.flat)
        ms_e, ms_d, ms_t = event_ms(lambda: hw.encode(xw), 3), event_ms(lambda: hw.decode(zw), 3), event_ms(lambda: hw.fwd_bwd(xw, gw), 2)
        enc_macs = F * 200 + 200 * 100 + 100 * 50 + 50 * Z
        flop_e, flop_t = 2 * enc_macs, 2 * (3 * 2 * enc_macs - F * 200)      # train: fwd + dW + dX (no dX for en1)
        res["wide_class"][f"ae_{F}_{Z}" + ("_class_forced" if forced else "")] = {
            "path": hw.path, "rows": rows, "encode_rows_per_s": rows / ms_e * 1e3, "decode_rows_per_s": rows / ms_d * 1e3,
            "train_fwd_bwd_rows_per_s": rows / ms_t * 1e3, "encode_frac_of_fp32_mfma_peak": flop_e * rows / ms_e / 1e9 / PEAK_TFLOPS["fp32"],
            "train_frac_of_fp32_mfma_peak": flop_t * rows / ms_t / 1e9 / PEAK_TFLOPS["fp32"]}
        hw.close()
        del xw, zw, gw
    return res


if __name__ == "__main__":
    main()
