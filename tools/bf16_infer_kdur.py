#!/usr/bin/env python3
"""Kernel durations of the bf16 encode of AE(24, 15) against the row count from a rocprofv3 kernel trace (host launch latency excluded):
    rocprofv3 --kernel-trace --output-format csv -d OUT -o run -- python3 tools/bf16_infer_kdur.py     then     ... kdur.py --read OUT
Row counts are whole rounds of the grid (131,072 rows = one 64-row pass for each of the 2,048 waves) and the benchmark's 1,000,000."""
import sys, os, glob, csv, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
NS = [16384, 65536, 131072, 262144, 524288, 1000000, 1048576, 2097152, 4194304]
if len(sys.argv) > 2 and sys.argv[1] == "--read":
    rows = []
    for f in glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows = [r for r in rows if "bf16_infer_kernel" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    per = 2 * 30                                       # (f64 rows, f32 rows) x 30 launches per size
    for k, n in enumerate(NS):
        for j, tag in enumerate(("float64 rows", "float32 rows")):
            d = sorted(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[k * per + 30 * j + 10:k * per + 30 * j + 30])
            print(f"{n:9d} {tag}: median {d[len(d) // 2] / 1e3:7.1f} us  min {d[0] / 1e3:7.1f} us grid {rows[k * per + 30 * j].get('Grid_Size_X', rows[k * per + 30 * j].get('Grid_Size', '?'))}")
    sys.exit(0)
import numpy as np, torch
from baler_amd import native
from oracle import c_oracle as orc
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from _gpu_warm import warm
dims = orc.ae_dims(24, 15)
h = native.Handle(dims, "bf16")
h.load_params(torch.from_numpy(np.concatenate([orc.formula_params(dims, 1), [0.0]]).astype(np.float32)).cuda())
warm(150.0)
for n in NS:
    for din in (torch.float64, torch.float32):
        x = torch.rand((n, 24), dtype=din, device="cuda")
        o = torch.empty((n, 15), dtype=torch.float64, device="cuda")
        for _ in range(30):
            h.encode(x, out=o)
        torch.cuda.synchronize()
print("done")
