#!/usr/bin/env python3
"""Turn the rocprofv3 CSVs of tools/profile_r1.sh (gpurun_out/prof_f, pmcf_f, pmcf_w, pmcf_m, bench_final.json) into the
committed summaries profiles/r1_kernel_stats.csv, profiles/r1_pmc_summary.json, profiles/r1_bench.json and print the
numbers that profiles/README.md quotes."""
import collections
import csv
import glob
import json
import shutil


def load(pattern):
    d = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(f)):
            kn = r['Kernel_Name']
            k = ('train_dec_kernel' if 'train_dec' in kn else 'train_enc_kernel' if 'train_enc' in kn else
                 'lat2_chain_kernel' if 'lat2_chain' in kn else 'lat2_dw_kernel' if 'lat2_dw' in kn else 'reduce_slabs_k' if 'reduce_slabs' in kn else
                 'infer_kernel<encode>' if ('infer_kernel<24, 15, 0>' in kn or 'infer2_kernel<24, 15, 0>' in kn) else
                 'infer_kernel<decode>' if ('infer_kernel<24, 15, 1>' in kn or 'infer2_kernel<24, 15, 1>' in kn) else 'adam_k' if 'adam_k' in kn else None)
            if k:
                d[k][r['Counter_Name']].append(float(r['Counter_Value']))
    return {k: {c: (max(v) if k in ('reduce_slabs_k', 'adam_k') else sum(v) / len(v)) for c, v in cs.items()}
            for k, cs in d.items()}


f, w, m = (load(f'gpurun_out/pmcf_{x}/**/*counter_collection.csv') for x in 'fwm')
out = {"note": "per launch, 1,000,000 rows (lat2_* kernels: 512 rows), fp32 mode; FETCH_SIZE/WRITE_SIZE in KB as reported by "
               "rocprofv3 (separate --pmc passes, --kernel-trace only); hbm_bytes = (2*FETCH_SIZE + WRITE_SIZE)*1024 (gfx950 "
               "correction: FETCH_SIZE reports half of a wide coalesced read; checked on minmax_partial: 96 MB reported for a "
               "192 MB read)", "kernels": {}}
for k in sorted(set(f) | set(w)):
    fs, ws = f.get(k, {}).get('FETCH_SIZE', 0.0), w.get(k, {}).get('WRITE_SIZE', 0.0)
    out["kernels"][k] = {"FETCH_SIZE_KB": fs, "WRITE_SIZE_KB": ws, "hbm_bytes": (2 * fs + ws) * 1024}
    out["kernels"][k].update(m.get(k, {}))
out["fwd_bwd_hbm_bytes_per_launch"] = sum(out["kernels"][k]["hbm_bytes"] for k in ('train_dec_kernel', 'train_enc_kernel', 'reduce_slabs_k'))
out["rows"] = 1000000
json.dump(out, open('profiles/r1_pmc_summary.json', 'w'), indent=1)
shutil.copy(glob.glob('gpurun_out/prof_f/**/*kernel_stats.csv', recursive=True)[0], 'profiles/r1_kernel_stats.csv')
d = json.load(open('gpurun_out/bench_final.json'))
d["roofline"]["traffic"] = out["fwd_bwd_hbm_bytes_per_launch"]
json.dump(d, open('profiles/r1_bench.json', 'w'), indent=1)
k = out["kernels"]
for kn in ('train_dec_kernel', 'train_enc_kernel', 'infer_kernel<encode>'):
    v = k[kn]
    busy = v['SQ_VALU_MFMA_BUSY_CYCLES'] / (1024 * v['GRBM_GUI_ACTIVE'] / 8.0)
    print(kn, f"busy {100 * busy:.1f}%  valu/mfma {(v['SQ_INSTS_VALU'] - v['SQ_INSTS_MFMA']) / v['SQ_INSTS_MFMA']:.2f}  "
          f"wait_any {v['SQ_WAIT_ANY'] / v['SQ_WAVE_CYCLES']:.3f} fetch {v['FETCH_SIZE_KB'] / 1e3:.0f} MB write {v['WRITE_SIZE_KB'] / 1e3:.0f} MB "
          f"hbm {v['hbm_bytes'] / 1e6:.0f} MB mfma {v['SQ_INSTS_MFMA'] / 1e6:.1f} M")
print('reduce', {a: round(b) for a, b in k['reduce_slabs_k'].items() if 'SIZE' in a or 'bytes' in a})
print("fwd_bwd traffic MB", out["fwd_bwd_hbm_bytes_per_launch"] / 1e6)
for r in list(csv.DictReader(open('profiles/r1_kernel_stats.csv')))[:4]:
    print(r['Name'].split('(')[0][-42:], r['Calls'], f"{float(r['AverageNs']) / 1e6:.4f} ms")
print({kk: d[kk] for kk in ['value', 'ms_per_step', 'encode_rows_per_s', 'decode_rows_per_s', 'encode_tflops', 'train_bs512_us_per_step', 'train_bs512_rows_per_s']})
print(d['roofline'])
print(d['cpu_baseline'])
