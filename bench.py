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

# algorithmic (unpadded) work per row of AE(24,15): SURVEY.md section 8(d) / BASELINE.md section 4
FLOP_TRAIN_ROW = 357_000
FLOP_ENCODE_ROW = 61_100
PEAK_TFLOPS = {"fp32": 157.3, "fp64": 78.6, "bf16": 2500.0}
DTYPE_NAME = {"fp32": "f32", "fp64": "f64", "bf16": "bf16"}


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=30)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--rows", type=int, default=1_000_000, help="rows per GPU")
    ap.add_argument("--mode", default=os.environ.get("BALER_AMD_MODE", "fp32"), choices=["fp32", "fp64"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip encode/decode/bs512 side measurements")
    ap.add_argument("--cpu-rows", type=int, default=400_000)
    return ap.parse_args()


def log(msg):
    print(f"[bench +{time.perf_counter() - T0:7.1f}s] {msg}", file=sys.stderr, flush=True)


T0 = time.perf_counter()


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


def cpu_baseline(rows):
    """The reference's loops restated in plain PyTorch fp64 (oracle/torch_ref.py, bit-identical to the
    imported reference in the authoring container): training.fit with DataLoader + loss.item() per step,
    batch 512; helper.compress's encode loop.  Bounded sample of the same synthetic workload."""
    from baler_amd import synth
    from oracle import c_oracle as orc
    from oracle import torch_ref
    data = orc.normalize(synth.cms_rows(rows))
    model = torch_ref.load_flat(torch_ref.DenseAE(24, 15), orc.formula_params(orc.ae_dims(24, 15), 7))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3)
    # 512-row fp64 GEMMs do not scale to every hardware thread of a big host: give the CPU its best
    # intra-op thread count (short calibration), and report the count actually used.
    hw = os.cpu_count() or 1
    best, cores = None, 1
    for nt in sorted({min(hw, c) for c in (4, 8, 16, 32, 64)}):
        torch.set_num_threads(nt)
        cal = torch_ref.make_loader(data[:10240], 512)
        torch_ref.fit_epoch(model, opt, cal)
        t0 = time.perf_counter()
        torch_ref.fit_epoch(model, opt, cal)
        dtc = time.perf_counter() - t0
        log(f"cpu baseline calibration: {nt} threads -> {10240 / dtc:.0f} rows/s")
        if best is None or dtc < best:
            best, cores = dtc, nt
    torch.set_num_threads(cores)
    rows = int(min(rows, max(20480, 20.0 * 10240 / best)))  # bound the timed epoch to ~20 s
    data = data[:rows]
    dl = torch_ref.make_loader(data, 512)
    t0 = time.perf_counter()
    torch_ref.fit_epoch(model, opt, dl)
    t_train = time.perf_counter() - t0
    t0 = time.perf_counter()
    torch_ref.compress_loop(model, data[:rows // 2], 512)
    t_enc = time.perf_counter() - t0
    return {
        "value": rows / t_train, "unit": "rows/s", "cores": cores, "kind": "port",
        "encode_rows_per_s": (rows // 2) / t_enc,
        "sample": f"{rows} rows x 24 cols, 1 epoch of training.fit at batch_size 512 through DataLoader "
                  f"(fp64, torch {torch.__version__}, {cores} threads); encode loop of helper.compress on "
                  f"{rows // 2} rows",
    }


def main():
    a = parse()
    from baler_amd import dist as bdist
    from baler_amd import native, synth
    from baler_amd.modules import models

    rank, world, local = bdist.init_from_env()
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
        if world > 1:
            bdist.allreduce_sum(grads)
        state["t"] += 1
        h.adam_step(flat, grads, m, v, state["t"], 1e-3, loss_accum=loss_acc)

    log("data resident, model ready")
    for _ in range(a.warmup):
        train_step()
    torch.cuda.synchronize()
    log("warm-up done")
    dt = timed(lambda: train_step(True), a.steps, world, dev)
    rows_total = world * a.rows * a.steps
    value = rows_total / dt
    k_ms = float(np.mean([e0.elapsed_time(e1) for e0, e1 in ev]))
    log(f"train: {value:.4g} rows/s, {1e3 * dt / a.steps:.3f} ms/step, fwd_bwd {k_ms:.3f} ms")
    final_loss = float(grads[-1].item())

    out = {
        "metric": "rows/sec (train) + rows/sec (encode), CMS 24-col AE at 1/2/4/8 GPUs",
        "value": value, "unit": "rows/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
        "ms_per_step": 1e3 * dt / a.steps, "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None, "dtype": DTYPE_NAME[a.mode], "data": "synthetic",
        "config": {
            "workload": f"CMS 24-col AE(24,15) train step (fwd+loss+bwd+Adam), {a.rows} synthetic rows per GPU "
                        f"resident in HBM as float64, one optimizer step per pass (global batch = n_gpus x {a.rows}), "
                        f"{a.mode} MFMA, random-init weights",
            "rows_per_gpu": a.rows, "n_cols": 24, "latent": 15, "parallelism": f"dp{world}",
            "compute_mode": a.mode,
        },
        "train_rows_per_s": value, "last_batch_loss": final_loss,
        "rccl_ranks": bdist.rank_world()[1],
        "dist_backend": (torch.distributed.get_backend() if world > 1 else None),
    }

    achieved = FLOP_TRAIN_ROW * a.rows / (k_ms * 1e-3) / 1e12
    # HBM traffic per launch comes from rocprofv3 PMC passes of this same command (FETCH_SIZE / WRITE_SIZE in
    # separate runs, gfx950 x2 read correction), summarised under profiles/; counters cannot be read in-process
    traffic = None
    try:
        with open(os.path.join(REPO, "profiles", "r1_pmc_summary.json")) as f:
            pm = json.load(f)
        if pm.get("rows") == a.rows and a.mode == "fp32":
            traffic = pm["fwd_bwd_hbm_bytes_per_launch"]
    except (OSError, ValueError, KeyError):
        pass
    out["roofline"] = {
        "bound": "mfma",
        "kernel": "bamd_fwd_bwd = train_dec_kernel + train_enc_kernel (+ 0.02 ms partial-gradient reduction; the last 576 rows on the small-batch kernels)",
        "achieved": achieved, "peak": PEAK_TFLOPS[a.mode], "unit": "TFLOP/s",
        "frac": achieved / PEAK_TFLOPS[a.mode], "traffic": traffic,
        "launch_ms": k_ms, "algorithmic_flop_per_row": FLOP_TRAIN_ROW, "rows_per_launch": a.rows,
    }

    if not a.no_extras:
        # encode / decode passes over the same resident rows (no collectives: rows shard naturally)
        z = h.encode(x)
        t_enc = timed(lambda: h.encode(x), max(3, a.steps // 2), world, dev)
        n_enc = max(3, a.steps // 2)
        out["encode_rows_per_s"] = world * a.rows * n_enc / t_enc
        t_dec = timed(lambda: h.decode(z), n_enc, world, dev)
        out["decode_rows_per_s"] = world * a.rows * n_enc / t_dec
        out["encode_tflops"] = FLOP_ENCODE_ROW * a.rows * n_enc / t_enc / 1e12
        if a.mode == "fp32":
            # the bf16 inference mode on the same rows and weights (a throughput mode: ~2e-3 / 6e-3 rel. error on
            # encode / decode, not the 1e-5 parity mode that `value` is measured in)
            hb = native.Handle(model.dims, "bf16")
            hb.load_params(flat)
            zb = hb.encode(x)
            out["bf16_encode_rel_err_vs_fp32"] = float((zb.double() - z.double()).norm() / z.double().norm())
            t_b = timed(lambda: hb.encode(x), n_enc, world, dev)
            out["bf16_encode_rows_per_s"] = world * a.rows * n_enc / t_b
            t_b = timed(lambda: hb.decode(z), n_enc, world, dev)
            out["bf16_decode_rows_per_s"] = world * a.rows * n_enc / t_b
            hb.close()
        # strict reference batching: global batch 512 x n_gpus... kept at 512 rows per GPU, sequential steps
        nb = 400
        def bs512_pass():
            for i in range(nb):
                state["t"] += 1
                if world == 1:      # what training.fit issues: one bamd_train_step per batch
                    h.train_step(x[i * 512:(i + 1) * 512], flat, m, v, state["t"], 1e-3, loss_accum=loss_acc)
                    continue
                h.fwd_bwd(x[i * 512:(i + 1) * 512], grads)
                bdist.allreduce_sum(grads)
                h.adam_step(flat, grads, m, v, state["t"], 1e-3, loss_accum=loss_acc)
        bs512_pass()
        t512 = timed(bs512_pass, 1, world, dev)
        out["train_bs512_rows_per_s"] = world * 512 * nb / t512
        out["train_bs512_us_per_step"] = 1e6 * t512 / nb
        log(f"encode {out['encode_rows_per_s']:.4g} rows/s, decode {out['decode_rows_per_s']:.4g} rows/s, "
            f"bs512 {out['train_bs512_rows_per_s']:.4g} rows/s ({out['train_bs512_us_per_step']:.1f} us/step)")

    if rank == 0 and world == 1 and not a.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(a.cpu_rows)

    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        import torch.distributed as td
        td.barrier()
        td.destroy_process_group()


if __name__ == "__main__":
    main()
