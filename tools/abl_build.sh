#!/bin/bash
# Build an A/B variant of ONE kernel source into .abl/NAME.so (the other objects come from the regular build):
#   tools/abl_build.sh NAME SRC.hip [-DFLAG ...]        e.g. tools/abl_build.sh dw4 bf16_train.hip -DBAMD_BF16_DWDEPTH=4
# SRC may also be a path outside csrc named <object>__<tag>.hip (an older revision of that file); run with BALER_AMD_LIB=$PWD/.abl/NAME.so
set -e
R=$(cd "$(dirname "$0")/.." && pwd)
NAME=$1; SRC=$2; shift 2
C=$R/baler_amd/csrc
mkdir -p $R/.abl /tmp/abl
BASE=$(basename $SRC .hip)
[ -f "$SRC" ] || SRC=$C/$SRC
EXTRA=""
case ${BASE%%__*} in fused*|bf16_train*) EXTRA="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -Wno-unused-function $EXTRA -I$C -I$R/include "$@" -c $SRC -o /tmp/abl/$NAME.o
OBJS=""
for o in api elementwise generic fused fused64 fused64q fused64i fused64j swd bf16 bf16_train comm; do
  if [ "$o" = "${BASE%%__*}" ]; then OBJS="$OBJS /tmp/abl/$NAME.o"; else OBJS="$OBJS $C/$o.o"; fi
done
hipcc --offload-arch=gfx950 -shared -fPIC -o $R/.abl/$NAME.so $OBJS
echo built .abl/$NAME.so
