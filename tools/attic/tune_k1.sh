#!/bin/bash
# On the GPU box: K1 with other pixels per lane / prefetch depths / waves per SIMD, with and without its stores (tuning aid).
# Every variant is a `make tuning` build (-DA3_TUNING + the variant's -D flags) in build/tuning/, loaded through A3_HIP_LIB:
# the product library aruco3_amd/libaruco3_hip.so is never touched.  CFGS="lpx pf waves ..."
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
LIB=$ROOT/build/tuning/libaruco3_hip.so
B="python3 $ROOT/bench.py --device-synth --no-cpu-baseline --no-other-workloads --repeats 5 --steps 20 --warmup 3"
show() { python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$1', 'k1_ms', d['stage_ms_per_step']['threshold'], 'fps', d['value'], d['frames_with_all_ids_correct'])"; }
for rep in 1 2; do
for cfg in ${CFGS:-"8 3 3" "16 3 2" "8 5 3" "8 3 4"}; do
  set -- ${cfg//_/ }   # "16 3 2" or 16_3_2
  make -C $ROOT/aruco3_amd/csrc tuning TUNE_FLAGS="-DA3_T_LPX=$1 -DA3_T_PF=$2 -DA3_T_WAVES=$3" > /dev/null 2>&1 || exit 1
  for fl in ${FLUSHES:--1 128}; do (cd $ROOT && A3_HIP_LIB=$LIB A3_K1_FLUSH=$fl $B 2>/dev/null | tail -1 | show "LPX=$1 PF=$2 waves=$3 flush=$fl"); done
done; done
