#!/bin/bash
# Round-5 evidence run on the GPU box (through gpurun): the bench line, rocprofv3 kernel stats of the same command and the counter
# passes (separate --pmc runs, --kernel-trace only) that tools/summarize_r5.py folds into profiles/.
#   gpurun --timeout 3000 -- 'bash tools/profile_r5.sh'
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r5p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="timeout 400"
python3 $R/bench.py > $O/bench.json 2> $O/bench.log
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
PM="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES"
prof() {   # prof TAG COMMAND...: kernel stats + FETCH / WRITE / SQ counter passes
    local tag=$1; shift
    $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}stats -o run -- "$@" > $O/${tag}stats.log 2>&1
    $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}pmc_f -o run -- "$@" > $O/${tag}pmc_f.log 2>&1
    $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${tag}pmc_w -o run -- "$@" > $O/${tag}pmc_w.log 2>&1
    $T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/${tag}pmc_m -o run -- "$@" > $O/${tag}pmc_m.log 2>&1
}
prof "" $B
prof b python3 $R/tools/bench_bf16_train.py                      # the shipped bf16 training pair (BASELINE configs[1] names bf16)
export BALER_AMD_BF16_TRAIN_V2=1                                  # (an environment variable of THIS shell: the profiled program is still python3 itself)
prof r python3 $R/tools/bench_bf16_train.py                      # the round-5 register-chain pair
$T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS --output-format csv -d $O/rpmc_l -o run -- python3 $R/tools/bench_bf16_train.py > $O/rpmc_l.log 2>&1
export BALER_AMD_BF16_TRAIN_V2=3                                  # the four launches with eight waves per workgroup (two waves per SIMD)
prof q python3 $R/tools/bench_bf16_train.py
unset BALER_AMD_BF16_TRAIN_V2
prof c python3 $R/tools/bench_c4.py 32768                         # CFD_dense_AE(2500, 25), exact instantiation
prof k python3 $R/tools/prof_wide_class.py                        # the run-time-width wide class on CFD_dense_AE(900, 9)
prof s python3 $R/tools/bench_one_batch.py 512 400                # the reference's own regime: 512-row optimiser steps
prof f python3 $R/tools/prof_fp64.py                              # the fp64 kernels (262,144-row launches; 512-row steps)
prof i python3 $R/tools/bench_bf16_infer.py                       # bf16 inference of the 24-column model (scalar LeakyReLU multiplies)
# per-phase shader-clock timeline of the register-chain pair (a -DBAMD_BF16_TRACE build in .abl/) and the instruction-cost probe
if [ -f $R/.abl/btrace.so ]; then BALER_AMD_LIB=$R/.abl/btrace.so timeout 200 python3 $R/tools/bf16_trace2.py > $O/regchain_trace.txt 2>&1; fi
if [ -x $R/tools/probe/valu_beside_mfma_probe.out ]; then timeout 120 $R/tools/probe/valu_beside_mfma_probe.out > $O/valu_probe.txt 2>&1; fi
timeout 300 python3 $R/tools/bench_wide_class.py > $O/wide_class_bench.txt 2>&1
timeout 200 python3 $R/tools/bench_mid_width_train.py > $O/mid_width_train.txt 2>&1
timeout 200 python3 $R/tools/bench_mid_width_wide.py > $O/mid_width_wide.txt 2>&1
timeout 60 $R/tools/probe/launch_gap_probe.out > $O/launch_gap_probe.txt 2>&1
for d in stats bstats rstats qstats cstats kstats sstats fstats istats; do echo "== $d"; python3 $R/tools/kstats.py $O/$d 8; done
tail -1 $O/bench.json | cut -c1-400
