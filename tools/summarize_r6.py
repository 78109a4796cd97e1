#!/usr/bin/env python3
"""Fold the rocprofv3 CSVs of tools/profile_r6.sh (gpurun_out/r6p) into the committed summaries
profiles/r6_*_kernel_stats.csv, profiles/r6_bench.json and profiles/pmc_summary.json (the file
bench.py quotes `roofline.traffic` from -- stamped with the hash of the kernel sources it was measured on)."""
import collections
import csv
import glob
import json
import os
import shutil
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
R = os.path.join(REPO, "gpurun_out", "r6p")

KEYS = (("train_dec_kernel", "train_dec"), ("train_enc_kernel", "train_enc"), ("reduce_slabs_k", "reduce_slabs"),
        ("lat2_chain_kernel", "lat2_chain"), ("lat4_chain_kernel", "lat4_chain"), ("lat2_dw_kernel", "lat2_dw"), ("adam_k", "adam_k"),
        ("infer64_kernel<ENCODE>", "infer64_kernel<24, 15, 0>"), ("infer64_kernel<DECODE>", "infer64_kernel<24, 15, 1>"),
        ("infer64_kernel<FORWARD>", "infer64_kernel<24, 15, 2>"), ("chain64q_kernel (4 rows per workgroup)", "chain64q_kernel"), ("chain64_kernel", "chain64_kernel<"), ("chain64r_kernel", "chain64r_kernel"), ("dw64m_kernel", "dw64m_kernel"), ("dw64x_kernel", "dw64x_kernel"), ("dw64_kernel", "dw64_kernel"),
        ("bf16_train_kernel<PART 0>", "bf16_train_kernel<24, 15, 0>"), ("bf16_train_kernel<PART 1>", "bf16_train_kernel<24, 15, 1>"),
        ("reduce_tiles_k", "reduce_tiles_k"),
        ("wide class: wide_encode_lds_kernel<4096, 15, WRT>", "wide_encode_lds_kernel<4096, 15"), ("wide class: wide_decode_lds_kernel<4096, 15, WRT>", "wide_decode_lds_kernel<4096, 15"),
        ("wide class: wide_train_fwd_kernel<4096, 15, WRT>", "wide_train_fwd_kernel<4096, 15"), ("wide class: wide_train_bwd_kernel<4096, 15, WRT>", "wide_train_bwd_kernel<4096, 15"),
        ("wide_encode_lds_kernel", "wide_encode_lds_kernel<2500, 25"), ("wide_infer_kernel<DECODE>", "wide_infer_kernel<2500, 25, 1"), ("wide_decode_lds_kernel", "wide_decode_lds_kernel<2500, 25"),
        ("wide_train_fwd_kernel", "wide_train_fwd_kernel<2500, 25, true>"), ("wide_train_bwd_kernel", "wide_train_bwd_kernel<2500, 25>"),
        ("wide_bf16_train_fwd_kernel", "wide_bf16_train_fwd_kernel<2500, 25"), ("wide_bf16_train_bwd_kernel", "wide_bf16_train_bwd_kernel<2500, 25"),
        ("dw_wide_bf16_k<P = dZ>", "dw_wide_bf16_k<true"), ("dw_wide_bf16_k<P = [X|1]>", "dw_wide_bf16_k<false"),
        ("wide_bf16_encode_dma_kernel", "wide_bf16_encode_dma_kernel<2500, 25"), ("C5: wide_bf16_encode_dma_kernel<512, 6>", "wide_bf16_encode_dma_kernel<512, 6"), ("wide_bf16_decode_kernel", "wide_bf16_decode_kernel<2500, 25"),
        ("bf16_infer_kernel<encode>", "bf16_infer_kernel<24, 15, false"), ("bf16_infer_kernel<decode>", "bf16_infer_kernel<24, 15, true"),
        ("dw_wide_k<P = dZ>", "dw_wide_k<true>"), ("dw_wide_k<P = [X|1]>", "dw_wide_k<false>"), ("reduce_layers_k", "reduce_layers_k"))


def load(pattern):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            k = next((name for name, sub in KEYS if sub in r["Kernel_Name"]), None)
            if k:
                d[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return {k: {c: sum(v) / len(v) for c, v in cs.items()} for k, cs in d.items()}


def merge(prefix):
    f, w, m = (load(f"{R}/{prefix}pmc_{x}/**/*counter_collection.csv") for x in "fwm")
    out = {}
    for k in sorted(set(f) | set(w) | set(m)):
        fs, ws = f.get(k, {}).get("FETCH_SIZE", 0.0), w.get(k, {}).get("WRITE_SIZE", 0.0)
        e = {"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_bytes": (2 * fs + ws) * 1024}
        e.update(m.get(k, {}))
        if e.get("GRBM_GUI_ACTIVE"):
            e["mfma_busy"] = e.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / (1024 * e["GRBM_GUI_ACTIVE"] / 8)
        if e.get("SQ_INSTS_MFMA"):
            e["valu_per_mfma"] = (e["SQ_INSTS_VALU"] - e["SQ_INSTS_MFMA"]) / e["SQ_INSTS_MFMA"]
        if e.get("SQ_WAVE_CYCLES"):
            e["wait_any_frac"] = e.get("SQ_WAIT_ANY", 0) / e["SQ_WAVE_CYCLES"]
        out[k] = e
    return out


import bench  # noqa: E402  (source_hash)

fp32 = merge("")
bf16 = merge("b")
c4 = merge("c")
q64 = merge("q")
wclass = merge("k")
small = merge("s")
infer16 = merge("i")
f64 = merge("f")
out = {
    "note": "per launch, 1,000,000 rows, averages over the launches of `python3 bench.py --no-cpu-baseline --no-extras` (fp32) and "
            "`python3 tools/bench_bf16_train.py` (bf16); FETCH_SIZE / WRITE_SIZE in KB as rocprofv3 reports them (separate --pmc "
            "passes, --kernel-trace only); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950: FETCH_SIZE reports half of a wide "
            "coalesced read, MI355X_MICROARCH.md; checked on minmax_partial: 96 MB reported for a 192 MB read)",
    "source_hash": bench.source_hash(), "rows": 1000000, "kernels": fp32, "bf16_kernels": bf16,
    "c4_note": "CFD_dense_AE(2500, 25), 32768 frames per launch, fp32 and bf16 handles, 100 launches per entry point behind a 150-ms clock warm-up (6 of them warm-up), `python3 tools/prof_c4_r6.py 32768`", "c4_kernels": c4,
    "fp64_bs512_note": "512-row bamd_train_step of an fp64 handle (chain64q_kernel: four rows per workgroup on v_mfma_f64_4x4x4, + dw64_kernel<adam>), `python3 tools/prof_fp64_bs512.py`, 300 steps", "fp64_bs512_kernels": q64,
    "wide_class_note": "the run-time-width wide class on CFD_dense_AE(900, 9), 131,072 float32 rows per launch, 100 launches per entry point behind a 150-ms clock warm-up, `python3 tools/prof_wide_class.py`", "wide_class_kernels": wclass,
    "bf16_infer_note": "bf16 encode / decode of AE(24, 15) at 1M float64 rows, 4M float64 and 4M float32 rows (100 launches each behind a 150-ms clock warm-up: the counters are averages over all three), and the C5 bf16 encode at 262,144 rows, `python3 tools/prof_bf16_infer_r6.py`", "bf16_infer_kernels": infer16,
    "bs512_note": "512-row bamd_train_step, `python3 tools/bench_one_batch.py 512 400`", "bs512_kernels": small,
    "fp64_note": "fp64 handle, 262,144 rows per launch (512 rows for chain64 / dw64; dw64_kernel averages the finishing launches of both), `python3 tools/prof_fp64.py`", "fp64_kernels": f64,
    "fwd_bwd_hbm_bytes_per_launch": sum(fp32[k]["hbm_bytes"] for k in ("train_dec_kernel", "train_enc_kernel", "reduce_slabs_k") if k in fp32),
    "bf16_fwd_bwd_hbm_bytes_per_launch": sum(bf16[k]["hbm_bytes"] for k in ("bf16_train_kernel<PART 0>", "bf16_train_kernel<PART 1>", "reduce_tiles_k") if k in bf16),
}
json.dump(out, open(os.path.join(REPO, "profiles", "pmc_summary.json"), "w"), indent=1)
STATS = (("stats", "r6_kernel_stats.csv"), ("bstats", "r6_bf16_kernel_stats.csv"),
         ("cstats", "r6_c4_kernel_stats.csv"), ("kstats", "r6_wide_class_kernel_stats.csv"),
         ("sstats", "r6_bs512_kernel_stats.csv"), ("fstats", "r6_fp64_kernel_stats.csv"), ("qstats", "r6_fp64_bs512_kernel_stats.csv"),
         ("istats", "r6_bf16_infer_kernel_stats.csv"))
for src, dst in (("wide_class_bench.txt", "r6_wide_class_bench.txt"), ("fp64_small_steps.txt", "r6_fp64_small_steps.txt")):
    if os.path.exists(f"{R}/{src}"):
        shutil.copy(f"{R}/{src}", os.path.join(REPO, "profiles", dst))
for d, name in STATS:
    g = glob.glob(f"{R}/{d}/**/*kernel_stats.csv", recursive=True)
    if g:
        shutil.copy(g[0], os.path.join(REPO, "profiles", name))
d = json.loads(open(f"{R}/bench.json").read().strip().splitlines()[-1])
if d.get("source_hash") == out["source_hash"]:
    d["roofline"]["traffic"] = out["fwd_bwd_hbm_bytes_per_launch"]
json.dump(d, open(os.path.join(REPO, "profiles", "r6_bench.json"), "w"), indent=1)
if os.path.exists(f"{R}/bench_pg.json"):
    lines = [l for l in open(f"{R}/bench_pg.json").read().strip().splitlines() if l.startswith("{")]
    if lines:
        json.dump(json.loads(lines[-1]), open(os.path.join(REPO, "profiles", "r6_bench_forced_pg.json"), "w"), indent=1)
for name, tab in (("fp32", fp32), ("bf16", bf16), ("c4", c4), ("wide class", wclass), ("bs512", small), ("fp64", f64), ("fp64 bs512", q64), ("bf16 infer", infer16)):
    for k, v in tab.items():
        print(f"{name} {k:28s} busy {100 * v.get('mfma_busy', 0):5.1f}%  valu/mfma {v.get('valu_per_mfma', 0):.2f}  wait_any {v.get('wait_any_frac', 0):.3f}  "
              f"hbm {v.get('hbm_bytes', 0) / 1e6:7.1f} MB  mfma {v.get('SQ_INSTS_MFMA', 0) / 1e6:.1f} M  lds_conflict {v.get('lds_conflict_frac', 0):.2f}")
print("fp32 fwd_bwd traffic MB", out["fwd_bwd_hbm_bytes_per_launch"] / 1e6, " bf16", out["bf16_fwd_bwd_hbm_bytes_per_launch"] / 1e6)
for f in [name for _, name in STATS if os.path.exists(os.path.join(REPO, "profiles", name))]:
    print("--", f)
    for r in list(csv.DictReader(open(os.path.join(REPO, "profiles", f))))[:7]:
        print(r["Name"].split("(")[0][-46:], r["Calls"], f"{float(r['AverageNs']) / 1e6:.4f} ms")
print({k: d.get(k) for k in ("value", "ms_per_step", "encode_rows_per_s", "decode_rows_per_s", "bf16_train_rows_per_s", "train_bs512_us_per_step")})
print(d["roofline"])
