"""Per-kernel averages of the rocprofv3 counter passes written by tools/prof.sh:  python tools/pmcsum.py gpurun_out/NAME [substr]"""
import collections
import csv
import glob
import sys

root = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
d = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(root + "/pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        d[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(d, key=lambda k: -sum(d[k].get("SQ_WAVE_CYCLES", [0]))):
    if want not in k:
        continue
    c = {n: sum(v) / len(v) for n, v in d[k].items()}
    name = k.split("(")[0][-70:]
    out = [f"{name:70s} n={len(next(iter(d[k].values())))}"]
    if c.get("GRBM_GUI_ACTIVE"):
        out.append(f"mfma_busy {c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) / (1024 * c['GRBM_GUI_ACTIVE'] / 8):.3f}")
    if c.get("SQ_INSTS_MFMA"):
        out.append(f"valu/mfma {(c['SQ_INSTS_VALU'] - c['SQ_INSTS_MFMA']) / c['SQ_INSTS_MFMA']:.2f} mfma {c['SQ_INSTS_MFMA'] / 1e6:.2f}M")
    if c.get("GRBM_GUI_ACTIVE"):
        out.append(f"gui_cyc {c['GRBM_GUI_ACTIVE'] / 8 / 1e3:.1f}k wave_cyc {4 * c.get('SQ_WAVE_CYCLES', 0) / 1e6:.1f}M")
    if c.get("SQ_WAVE_CYCLES"):
        out.append(f"wait_any {c.get('SQ_WAIT_ANY', 0) / c['SQ_WAVE_CYCLES']:.3f}")
    if c.get("SQ_LDS_IDX_ACTIVE"):
        out.append(f"lds_conflict/active {c.get('SQ_LDS_BANK_CONFLICT', 0) / c['SQ_LDS_IDX_ACTIVE']:.3f} lds_insts {c.get('SQ_INSTS_LDS', 0) / 1e6:.2f}M "
                   f"lds_active_cyc {c['SQ_LDS_IDX_ACTIVE'] / 1e6:.1f}M wait_inst_any {c.get('SQ_WAIT_INST_ANY', 0) / 1e6:.1f}M "
                   f"active_inst {c.get('SQ_ACTIVE_INST_ANY', 0) / 1e6:.1f}M wait_inst_lds {c.get('SQ_WAIT_INST_LDS', 0) / 1e6:.1f}M")
    if "FETCH_SIZE" in c or "WRITE_SIZE" in c:
        out.append(f"fetch {c.get('FETCH_SIZE', 0) / 1e3:.1f} MB write {c.get('WRITE_SIZE', 0) / 1e3:.1f} MB hbm {(2 * c.get('FETCH_SIZE', 0) + c.get('WRITE_SIZE', 0)) / 1e3:.1f} MB")
    print("  ".join(out))
