#!/bin/bash
# On the GPU box: tools/kernel_probe.py rows matching PATTERN for builds under build/<name>/ in alternation (a3_debug_kernel_time).
#   VARIANTS="da_old da_new" PATTERN="dart_assign" REPS=2 tools/ab_probe.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd "$ROOT"
for rep in $(seq 1 ${REPS:-2}); do
for v in ${VARIANTS:-da_old da_new}; do
  lib=$ROOT/build/$v/libaruco3_hip.so; [ "$v" = product ] && lib=$ROOT/aruco3_amd/libaruco3_hip.so
  echo "== $v"
  A3_HIP_LIB=$lib timeout -k 10 300 python3 tools/kernel_probe.py 2>/dev/null | grep -E "${PATTERN:-dart_assign}" | tr '\n' ';'; echo
done
done
