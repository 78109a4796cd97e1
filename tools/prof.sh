#!/bin/bash
# rocprofv3 wrapper for the GPU box:  gpurun -- 'bash tools/prof.sh NAME [pmc] -- python3 tools/x.py ...'
#   NAME        output directory gpurun_out/NAME
#   pmc         also run the three counter passes (FETCH_SIZE / WRITE_SIZE / SQ set), --kernel-trace only
# The program after -- is started directly under rocprofv3 (no env / shell hop).  Every pass is bounded by `timeout`.
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}
NAME=$1; shift
PMC=0
if [ "$1" = "pmc" ]; then PMC=1; shift; fi
shift   # the --
O=$R/gpurun_out/$NAME
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
T=${PROF_TIMEOUT:-240}
timeout $T rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- "$@" > $O/stats.log 2>&1
python3 $R/tools/kstats.py $O/stats 14
if [ $PMC = 1 ]; then
  timeout $T rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_f -o run -- "$@" > $O/pmc_f.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_w -o run -- "$@" > $O/pmc_w.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_INSTS_VALU SQ_INSTS_MFMA SQ_WAIT_ANY SQ_WAVE_CYCLES --output-format csv -d $O/pmc_m -o run -- "$@" > $O/pmc_m.log 2>&1
  timeout $T rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD --output-format csv -d $O/pmc_l -o run -- "$@" > $O/pmc_l.log 2>&1
  python3 $R/tools/pmcsum.py $O
fi
