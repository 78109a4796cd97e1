#!/bin/bash
# Round-4 evidence run on the GPU box (through gpurun): the bench line, rocprofv3 kernel stats of the same command and the
# counter passes (separate --pmc runs, --kernel-trace only) that tools/summarize_r4.py folds into profiles/.
#   gpurun --timeout 2400 -- 'bash tools/profile_r4.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r4p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="timeout 400"
python3 $R/bench.py > $O/bench.json 2> $O/bench.log
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
PM="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/stats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o run -- $B > $O/pmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o run -- $B > $O/pmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/pmc_m -o run -- $B > $O/pmc_m.log 2>&1
# bf16 training kernels (BASELINE configs[1] names bf16)
X="python3 $R/tools/bench_bf16_train.py"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -o run -- $X > $O/bstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bpmc_f -o run -- $X > $O/bpmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bpmc_w -o run -- $X > $O/bpmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/bpmc_m -o run -- $X > $O/bpmc_m.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/bpmc_l -o run -- $X > $O/bpmc_l.log 2>&1
# CFD_dense_AE(2500, 25) (BASELINE configs[3]): wide-layer encode / decode / training kernels, 32768 frames
C="python3 $R/tools/bench_c4.py 32768"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/cstats -o run -- $C > $O/cstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cpmc_f -o run -- $C > $O/cpmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cpmc_w -o run -- $C > $O/cpmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/cpmc_m -o run -- $C > $O/cpmc_m.log 2>&1
# the same model on a BAMD_MODE_BF16 handle: training passes with the wide products on the bf16 MFMA
CB="python3 $R/tools/prof_wide_bf16_train.py"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/cbstats -o run -- $CB > $O/cbstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cbpmc_f -o run -- $CB > $O/cbpmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cbpmc_w -o run -- $CB > $O/cbpmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/cbpmc_m -o run -- $CB > $O/cbpmc_m.log 2>&1
# the reference's own regime: 512-row optimiser steps (lat4_chain_kernel + lat2_dw_kernel), and 4096-row steps (lat2_chain_kernel)
S="python3 $R/tools/bench_one_batch.py 512 400"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/sstats -o run -- $S > $O/sstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/spmc_f -o run -- $S > $O/spmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/spmc_w -o run -- $S > $O/spmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/spmc_m -o run -- $S > $O/spmc_m.log 2>&1
S4="python3 $R/tools/bench_one_batch.py 4096 200"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/mstats -o run -- $S4 > $O/mstats.log 2>&1
# fp64 (the reference's own dtype): register-chained inference kernel, fused training pair, 512-row step
F="python3 $R/tools/prof_fp64.py"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/fstats -o run -- $F > $O/fstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/fpmc_m -o run -- $F > $O/fpmc_m.log 2>&1
# bf16 inference (24-column model): kernel stats + the byte counters the review asked for
I="python3 $R/tools/bench_bf16_infer.py"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/istats -o run -- $I > $O/istats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/ipmc_f -o run -- $I > $O/ipmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/ipmc_w -o run -- $I > $O/ipmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/ipmc_m -o run -- $I > $O/ipmc_m.log 2>&1
# bf16 encode / decode of CFD_dense_AE(2500, 25) (the HBM-bound config of SURVEY 8(d)): loader-wave kernels, 131072 frames
W="python3 $R/tools/bench_c4_bf16.py 131072"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/wstats -o run -- $W > $O/wstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/wpmc_f -o run -- $W > $O/wpmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/wpmc_w -o run -- $W > $O/wpmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/wpmc_m -o run -- $W > $O/wpmc_m.log 2>&1
for d in stats bstats cstats cbstats sstats mstats fstats istats wstats; do echo "== $d"; python3 $R/tools/kstats.py $O/$d 8; done
tail -1 $O/bench.json | cut -c1-400
