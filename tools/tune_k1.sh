#!/bin/bash
# On the GPU box: rebuild k_threshold with other pixels per lane / prefetch depths / waves per SIMD and time K1, with and without
# its stores (tuning aid).  CFGS="lpx pf waves ..."
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd $ROOT/aruco3_amd/csrc
B="python3 $ROOT/bench.py --device-synth --no-cpu-baseline --no-other-workloads --repeats 5 --steps 20 --warmup 3"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k1_ms', d['stage_ms_per_step']['threshold'], 'fps', d['value'], d['frames_with_all_ids_correct'])"; }
for rep in 1 2; do
for cfg in ${CFGS:-"8 3 3" "16 3 2" "8 5 3" "8 3 4"}; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DA3_T_LPX=$1 -DA3_T_PF=$2 -DA3_T_WAVES=$3 -c k_threshold.hip -o k_threshold.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaruco3_hip.so a3_api.o k_threshold.o k_contours.o k_decode.o k_synth.o || exit 1
  for fl in -1 128; do (cd $ROOT && A3_K1_FLUSH=$fl $B 2>/dev/null | tail -1 | show "LPX=$1 PF=$2 waves=$3 flush=$fl"); done
done; done
