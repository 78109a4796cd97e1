#!/usr/bin/env python3
"""Rewrite the round-5 section at the top of profiles/README.md from profiles/r5_bench.json and profiles/pmc_summary.json
(after tools/profile_r5.sh + tools/summarize_r5.py):  python tools/profiles_readme_r5.py"""
import json, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
b = json.load(open(f"{R}/profiles/r5_bench.json"))
d = json.load(open(f"{R}/profiles/pmc_summary.json"))
oc = b['other_configs']; nt = oc['narrow_tables']; wc = oc['wide_class']; c4 = oc['c4_cfd_dense_2500_25']
rx = b['roofline_extra']; tb = rx['train_bf16']; sb = c4['train_step_by_reference_batch_size']
k, bk, fk, sk = d['kernels'], d['bf16_kernels'], d['fp64_kernels'], d['bs512_kernels']
def pct(x): return f"{100*x:.1f} %"
def row(tab, name):
    e = tab[name]
    return f"{pct(e['mfma_busy'])} | {e['valu_per_mfma']:.2f} | {pct(e['wait_any_frac'])} | {e.get('hbm_bytes',0)/1e6:.0f} MB"
sec = f"""# profiles — round 5 (1×MI355X; 1,000,000 synthetic CMS rows resident in HBM)

All numbers from the GPU box via `gpurun` (`tools/profile_r5.sh`, `tools/summarize_r5.py`, this section by `tools/profiles_readme_r5.py`; raw
rocprofv3 CSVs are scratch, `gpurun_out/r5p`). Committed here: `r5_bench.json` (the bench line), `rocprofv3 --kernel-trace --stats` summaries
`r5_kernel_stats.csv` (of `python3 bench.py --no-cpu-baseline --no-extras`), `r5_bf16_kernel_stats.csv` (`tools/bench_bf16_train.py`, the shipped
pair), `r5_bf16_regchain_kernel_stats.csv` / `r5_bf16_quad_kernel_stats.csv` (the two round-5 rewrites of the bf16 training pass,
`BALER_AMD_BF16_TRAIN_V2=1` / `3`), `r5_c4_kernel_stats.csv` (`tools/bench_c4.py 32768`), `r5_wide_class_kernel_stats.csv`
(`tools/prof_wide_class.py`: the run-time-width class on `CFD_dense_AE(900, 9)`), `r5_bs512_kernel_stats.csv` (`tools/bench_one_batch.py 512 400`),
`r5_fp64_kernel_stats.csv` (`tools/prof_fp64.py`), `r5_bf16_infer_kernel_stats.csv` (`tools/bench_bf16_infer.py`) and `pmc_summary.json` (counter
passes: FETCH_SIZE, WRITE_SIZE and the SQ counters each in their own `--pmc` run with `--kernel-trace` only; stamped with the hash of the kernel
sources, `{d['source_hash']}` = the build of this commit: `bench.py` quotes `roofline.traffic` from it only when the hash matches).
Measurements that are not profiles: `r5_fp64_dw_counters.txt` (L2 / fabric counters of the fp64 weight-gradient tile-block variants, DESIGN §4.8),
`r5_valu_beside_mfma_probe.txt` (what a VALU / LDS instruction costs beside an MFMA), `r5_bf16_regchain_trace.txt` (per-phase shader-clock
timeline of the register-chain pair), `r5_launch_gap_probe.txt` (launch floor of dependent kernels, stream vs graph), `r5_wide_class_bench.txt`
(class vs exact instantiation vs layer-wise), `r5_mid_width_train.txt` / `r5_mid_width_two_state.txt` (64..127-column tables).

| quantity (`r5_bench.json`, steady state: DESIGN.md §5) | round 5 | round 4 |
|---|---|---|
| fp32 train, one 1M-row step — `value` | **{b['value']/1e6:.1f} M rows/s**, {b['ms_per_step']:.3f} ms/step; `bamd_fwd_bwd` {b['roofline']['launch_ms']:.3f} ms by HIP events = {b['roofline']['achieved']:.1f} TFLOP/s = **{b['roofline']['frac']:.3f} of the fp32 MFMA peak** (kernels unchanged, 296–299 M box to box; closed-form ceiling 0.698, DESIGN §4.1); traffic {b['roofline']['traffic']/1e9:.2f} GB per launch | 298.5 M, 0.680 |
| fp32 encode / decode | {b['encode_rows_per_s']/1e9:.2f} / {b['decode_rows_per_s']/1e9:.2f} G rows/s | 1.93 / 1.97 |
| bf16 training (`roofline_extra.train_bf16`) | {b['bf16_train_rows_per_s']/1e9:.2f} G rows/s through the step; shipped pair {tb['launch_ms']:.3f} ms = **{tb['frac']:.3f} of the bf16 peak**; the two rewrites built this round: register-chain pair {tb['other_kernel_versions']['register_chain_pair']['launch_ms']:.3f} ms, four launches at two waves per SIMD {tb['other_kernel_versions']['quad_launches_two_waves_per_simd']['launch_ms']:.3f} ms — both slower, the pair stays (DESIGN §4.6) | 0.743 ms = 0.192 |
| bf16 encode / decode, 24 columns, float64 rows | {b['bf16_encode_rows_per_s']/1e9:.1f} / {b['bf16_decode_rows_per_s']/1e9:.1f} G rows/s | 10.3 / 8.9 |
| `batch_size = 512` step · by batch (fp32) | **{b['train_bs512_us_per_step']:.1f} µs** per call; ONE host call per epoch (`bamd_train_epoch`): {b['train_bs512_epoch_call']['us_per_step']:.2f} µs per step, host {b['train_bs512_epoch_call']['host_us_per_step']:.1f} µs · 4,096: {b['train_rows_per_s_by_batch']['4096']['us_per_step']:.1f} · 32,768: {b['train_rows_per_s_by_batch']['32768']['us_per_step']:.0f} · 262,144: {b['train_rows_per_s_by_batch']['262144']['us_per_step']:.0f} | 18.2 · 44.7 · 150 · 908 |
| **fp64** (`roofline_extra`) | encode {rx['encode_f64']['frac']:.2f} of the fp64 peak; **training at 1M rows {rx['train_f64']['launch_ms']:.2f} ms = {rx['train_f64']['frac']:.3f}** ({rx['train_f64']['rows_per_s']/1e6:.0f} M rows/s; exact weight-gradient tile blocks with two blocks of slices in flight: `dw64x_kernel` ~910 µs per 262,144 rows at {pct(fk['dw64x_kernel']['mfma_busy'])} MFMA busy, {2*fk['dw64x_kernel']['FETCH_SIZE_KB']*1024/1e9:.2f} GB fetched for 3.42 GB of images; round 4: `dw64m_kernel` 1,300 µs, 52.4 %, 5.1 GB); 65,536 rows {rx['train_f64_64k']['frac']:.3f}; 512-row step {rx['train_bs512_f64']['launch_us']:.1f} µs | 9.29 ms = 0.489 · 0.43 · 41.0 |
| C4 `CFD_dense_AE(2500,25)`, 32,768 frames | fp32 encode / decode / training {c4['encode_frac_of_mfma_peak']:.2f} / {c4['decode_frac_of_mfma_peak']:.2f} / {c4['train_frac_of_mfma_peak']:.2f} of the fp32 peak; bf16 encode / decode {c4['bf16_encode_frac_of_hbm']:.2f} / {c4['bf16_decode_frac_of_hbm']:.2f} of HBM, bf16 training {c4['bf16_train_vs_fp32']:.2f}× fp32 | same kernels |
| **C4 at the reference's own batch sizes** (`train_step_by_reference_batch_size`) | optimiser step of 60 rows **{sb['60']['us_per_step']:.0f} µs** (589 before this round), 6,000 rows {sb['6000']['us_per_step']:.0f} µs (698), 60 rows in float64 {sb['60_float64']['us_per_step']:.0f} µs (1,398); 1 row 94 µs (528), validation pass of 60 rows 40 µs (304) (`tools/bench_wide_small_step.py`, `bench_wide_small_validate.py`; DESIGN §4.7) | – |
| wide class `ImplWide<4096, 15/31/63, true>` (`other_configs.wide_class`, new) | AE(900,9): encode {wc['ae_900_9']['encode_frac_of_fp32_mfma_peak']:.2f}, training {wc['ae_900_9']['train_frac_of_fp32_mfma_peak']:.2f} of the fp32 peak · AE(1024,11): {wc['ae_1024_11']['encode_frac_of_fp32_mfma_peak']:.2f} / {wc['ae_1024_11']['train_frac_of_fp32_mfma_peak']:.2f} · AE(4096,41): {wc['ae_4096_41']['encode_rows_per_s']/1e6:.0f} M rows/s encode; class vs exact instantiation at 625 / 2500 columns within 4 % (`r5_wide_class_bench.txt`) | layer-wise (2–7× slower) |
| 64..127-column tables (`other_configs.narrow_tables`) | AE(64,16): encode {nt['ae_64_16']['encode_rows_per_s']/1e9:.2f} G rows/s, `fwd_bwd` {nt['ae_64_16']['train_fwd_bwd_rows_per_s']/1e6:.0f} M rows/s, 512-row step {nt['ae_64_16']['train_bs512_us_per_step']:.1f} µs · AE(80,16): {nt['ae_80_16']['encode_rows_per_s']/1e9:.2f} G / {nt['ae_80_16']['train_fwd_bwd_rows_per_s']/1e6:.0f} M / {nt['ae_80_16']['train_bs512_us_per_step']:.1f} µs | AE(64,16) 85 M `fwd_bwd`; AE(80,16) 0.46 G / 78 M / 420 µs |
| exafel `CFD_dense_AE(625, 7)`, 131,072 blocks | encode {oc['exafel_625_7']['encode_frac_of_mfma_peak']:.2f}, decode {oc['exafel_625_7']['decode_frac_of_mfma_peak']:.2f}, training {oc['exafel_625_7']['train_frac_of_mfma_peak']:.2f}; optimiser step of 32 rows 76 µs (224 before) | 0.77 / 0.79 / 0.63 |
| PCIe-inclusive, 10 M-row file | compress {b['pcie']['compress_rows_per_s']/1e6:.0f} M rows/s, decompress {b['pcie']['decompress_rows_per_s']/1e6:.0f} M; H2D {b['pcie']['h2d_gbs']:.0f} GB/s | 126 / 140 |
| CPU baseline (plain-PyTorch fp64 port of `training.fit`, bs 512, 1M rows) | {b['cpu_baseline']['value']/1e3:.0f} k rows/s on {b['cpu_baseline']['cores']} of 256 threads (the calibration's best) of an EPYC 9575F | 143 k |

## Counters (`pmc_summary.json`, round 5)

| kernel | MFMA busy | VALU per MFMA | `SQ_WAIT_ANY` | HBM bytes per launch (2·FETCH + WRITE) |
|---|---|---|---|---|
| `train_dec_kernel` (fp32, 1M rows) | {row(k,'train_dec_kernel')} |
| `train_enc_kernel` | {row(k,'train_enc_kernel')} |
| `bf16_train_kernel<PART 0>` (shipped) | {row(bk,'bf16_train_kernel<PART 0>')} |
| `bf16_train_kernel<PART 1>` | {row(bk,'bf16_train_kernel<PART 1>')} |
| **`dw64x_kernel`** (fp64 weight-gradient tile blocks, 262,144 rows) | {row(fk,'dw64x_kernel')} |
| `chain64r_kernel` (fp64 chain, 262,144 rows) | {row(fk,'chain64r_kernel')} |
| `lat4_chain_kernel` (512 rows) | {row(sk,'lat4_chain_kernel')} |
| `lat2_dw_kernel<adam>` (512 rows) | {row(sk,'lat2_dw_kernel')} |

The register-chain pair, the four-launch version, the wide class, the C4 kernels and the bf16 inference kernels: `regchain_kernels`,
`quad_kernels`, `wide_class_kernels`, `c4_kernels`, `bf16_infer_kernels` in the same file.

---

"""
p = f"{R}/profiles/README.md"
s = open(p).read()
mark = "# round 4 (kept for the history; superseded where round 5 re-measured)"
assert mark in s
open(p, "w").write(sec + s[s.index(mark):])
print("profiles/README.md: round-5 section rewritten")
