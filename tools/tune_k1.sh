#!/bin/bash
# Rebuild K1 with different compile-time knobs and time it inside bench.py (run on the GPU box).
set -u
cd "$(dirname "$0")/../aruco3_amd/csrc"
OUT=${1:-/tmp/tune_k1.log}
: > "$OUT"
for cfg in "3 3" "2 5" "4 3"; do
  set -- $cfg; w=$1; pf=$2
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -DA3_T_PF=$pf -DA3_T_WAVES=$w -c k_threshold.hip -o k_threshold.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaruco3_hip.so a3_api.o k_threshold.o k_contours.o k_decode.o || exit 1
  for rows in 106 256; do
    r=$(cd ../.. && A3_ROWS_PER_WAVE=$rows python bench.py --steps 8 --warmup 2 --frames-cache /tmp/c2frames --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stage_ms_per_step']['threshold'], d['roofline']['frac'], d['value'])")
    echo "waves=$w pf=$pf rows=$rows -> threshold_ms frac fps: $r" | tee -a "$OUT"
  done
done
