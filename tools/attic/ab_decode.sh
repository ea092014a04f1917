#!/bin/bash
# On the GPU box: k_decode alone (a3_debug_kernel_time through tools/kernel_probe.py) for builds under build/<name>/ in alternation.
#   VARIANTS="unpacked packed" REPS=2 tools/ab_decode.sh
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd "$ROOT"
for rep in $(seq 1 ${REPS:-2}); do
for v in ${VARIANTS:-unpacked packed}; do
  lib=$ROOT/build/$v/libaruco3_hip.so; [ "$v" = product ] && lib=$ROOT/aruco3_amd/libaruco3_hip.so
  echo "== $v"
  A3_HIP_LIB=$lib timeout -k 10 300 python3 tools/kernel_probe.py 256 decode 2>/dev/null | grep -E "^decode" | tr '\n' ';'; echo
done
done
