#!/bin/bash
# On the GPU box: builds of the library in alternation through tools/ab_streams.py (A3_HIP_LIB selects the copy).
#   VARIANTS="name:waves ..." (name = directory under build/, or "product"), SPECS_TMPL="shared:2:W:2,own:2:W:0" (W replaced)
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd "$ROOT"
for rep in $(seq 1 ${REPS:-1}); do
for v in ${VARIANTS:-product:2 rc3:3 rc2:2 lpx8:3}; do
  name=${v%%:*}; w=${v#*:}
  lib=$ROOT/build/$name/libaruco3_hip.so; [ "$name" = product ] && lib=$ROOT/aruco3_amd/libaruco3_hip.so
  echo "== $name (K1 waves per SIMD $w)"
  if [ "$name" != product ] && [ "$rep" = 1 ] && [ "${PARITY:-1}" = 1 ]; then
    A3_HIP_LIB=$lib timeout -k 10 300 python3 -m pytest tests/test_gpu_parity.py -x -q -k "threshold or golden or baseline_configs" 2>&1 | tail -2
  fi
  specs=$(echo "${SPECS_TMPL:-shared:2:W:2,own:2:W:0,own:3:W:0}" | sed "s/W/$w/g")
  A3_HIP_LIB=$lib timeout -k 10 300 python3 tools/ab_streams.py 256 40 ${ROUNDS:-3} $specs 2>&1 | grep -v amdgpu.ids
done
done
