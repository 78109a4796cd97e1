#!/bin/bash
# Round-1 evidence run on the GPU box (called through gpurun): bench line, rocprofv3 kernel stats and the three PMC
# passes that tools/summarize_profiles.py folds into profiles/.  Counters are collected with --kernel-trace only.
#   gpurun --timeout 2400 -- 'bash tools/profile_r1.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
python3 $R/bench.py > $O/bench_final.json 2> $O/bench_final.log
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_f -o runc -- $B > $O/prof_f.log 2>&1
X="python3 $R/bench.py --no-cpu-baseline"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmcf_f -o runc -- $X > $O/pmcf_f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmcf_w -o runc -- $X > $O/pmcf_w.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmcf_m -o runc -- $X > $O/pmcf_m.log 2>&1
ls $O/prof_f $O/pmcf_f $O/pmcf_w $O/pmcf_m
tail -1 $O/bench_final.json | cut -c1-300
