#!/bin/bash
# per-kernel times of the bf16 training pair under rocprofv3 (GPU box): tools/prof_bf16_train2.sh [outdir]
cd /tmp && export TMPDIR=/tmp
OUT=${1:-$GRAFT_REPO_ROOT/gpurun_out/prof_bf16}
rm -rf $OUT
rocprofv3 --kernel-trace --stats -d $OUT -o bf16 -- python3 $GRAFT_REPO_ROOT/tools/bench_bf16_train.py 1000000 30 > $OUT.log 2>&1
python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if float(r["Percentage"]) > 0.5:
            print(f'{r["Name"][:90]:90s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"])/1e3:9.1f} us  {r["Percentage"]}%')
PY
