#!/bin/bash
# Round-2 evidence run on the GPU box (through gpurun): the bench line, rocprofv3 kernel stats of the same command and
# the counter passes (separate --pmc runs, --kernel-trace only) that tools/summarize_r2.py folds into profiles/.
#   gpurun --timeout 2400 -- 'bash tools/profile_r2.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r2
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="timeout 400"
python3 $R/bench.py > $O/bench.json 2> $O/bench.log
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- $B > $O/stats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o run -- $B > $O/pmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o run -- $B > $O/pmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_m -o run -- $B > $O/pmc_m.log 2>&1
# bf16 training kernels (BASELINE configs[1] names bf16) and the bf16 inference kernels
X="python3 $R/tools/bench_bf16_train.py"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/bstats -o run -- $X > $O/bstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/bpmc_f -o run -- $X > $O/bpmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/bpmc_w -o run -- $X > $O/bpmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/bpmc_m -o run -- $X > $O/bpmc_m.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/bpmc_l -o run -- $X > $O/bpmc_l.log 2>&1
# CFD_dense_AE(2500, 25) (BASELINE configs[3]): wide-layer encode / decode / training kernels, 32768 frames
C="python3 $R/tools/bench_c4.py 32768"
$T rocprofv3 --kernel-trace --stats --output-format csv -d $O/cstats -o run -- $C > $O/cstats.log 2>&1
$T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/cpmc_f -o run -- $C > $O/cpmc_f.log 2>&1
$T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/cpmc_w -o run -- $C > $O/cpmc_w.log 2>&1
$T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/cpmc_m -o run -- $C > $O/cpmc_m.log 2>&1
python3 $R/tools/kstats.py $O/stats 8
python3 $R/tools/kstats.py $O/cstats 8
python3 $R/tools/kstats.py $O/bstats 6
tail -1 $O/bench.json | cut -c1-400
