#!/bin/bash
# Round-6 evidence run on the GPU box (through gpurun): the bench line, rocprofv3 kernel stats of the same command and the counter
# passes (separate --pmc runs, --kernel-trace only) that tools/summarize_r6.py folds into profiles/.
#   gpurun --timeout 3000 -- 'bash tools/profile_r6.sh'
# Every program is started directly under rocprofv3 (python3 itself, absolute script path: no env / shell hop), every pass bounded.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
O=$R/gpurun_out/r6p
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T="timeout 400"
if [ "${SKIP_BENCH:-0}" != "1" ]; then
python3 $R/bench.py > $O/bench.json 2> $O/bench.log
# the same program under a ONE-rank RCCL group (the rank is a child of torchrun, started before anything touches the GPU): dp_step et al.
BALER_AMD_FORCE_PG=1 timeout 900 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29513 \
    $R/bench.py --gpus 1 --no-cpu-baseline > $O/bench_pg.json 2> $O/bench_pg.log
fi
B="python3 $R/bench.py --no-cpu-baseline --no-extras"
PM="SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES"
prof() {   # prof TAG COMMAND...: kernel stats + FETCH / WRITE / SQ counter passes
    local tag=$1; shift
    $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/${tag}stats -o run -- "$@" > $O/${tag}stats.log 2>&1
    $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/${tag}pmc_f -o run -- "$@" > $O/${tag}pmc_f.log 2>&1
    $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/${tag}pmc_w -o run -- "$@" > $O/${tag}pmc_w.log 2>&1
    $T rocprofv3 --kernel-trace --pmc $PM --output-format csv -d $O/${tag}pmc_m -o run -- "$@" > $O/${tag}pmc_m.log 2>&1
}
LEGS=${LEGS:-"main b c k s f q i"}
has() { case " $LEGS " in *" $1 "*) return 0;; esac; return 1; }
if has main; then prof "" $B; fi
# the bf16 training pair (BASELINE configs[1] names bf16)
if has b; then prof b python3 $R/tools/bench_bf16_train.py 1000000 40; fi
# CFD_dense_AE(2500, 25): fp32 and bf16 handles, 100 launches per entry point
if has c; then prof c python3 $R/tools/prof_c4_r6.py 32768; fi
# the run-time-width wide class on CFD_dense_AE(900, 9), 100 launches each
if has k; then prof k python3 $R/tools/prof_wide_class.py; fi
# the reference's own regime in fp32: 512-row optimiser steps
if has s; then prof s python3 $R/tools/bench_one_batch.py 512 400; fi
# the fp64 kernels (262,144-row launches)
if has f; then prof f python3 $R/tools/prof_fp64.py; fi
# the reference's own regime in its own dtype: 512-row fp64 steps (4-row chain)
if has q; then prof q python3 $R/tools/prof_fp64_bs512.py; fi
# bf16 inference: 24 columns at 1M / 4M rows, C5
if has i; then prof i python3 $R/tools/prof_bf16_infer_r6.py; fi
timeout 300 python3 $R/tools/bench_wide_class.py > $O/wide_class_bench.txt 2>&1
timeout 120 python3 $R/tools/bench_fp64_small_steps.py > $O/fp64_small_steps.txt 2>&1
for d in stats bstats cstats kstats sstats fstats qstats istats; do echo "== $d"; python3 $R/tools/kstats.py $O/$d 8; done
tail -1 $O/bench.json | cut -c1-400
