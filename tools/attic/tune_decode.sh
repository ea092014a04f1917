#!/bin/bash
# On the GPU box: k_decode with other (samples in flight per lane, waves per SIMD, threads, sample order) -- `make tuning` builds in
# build/tuning/, loaded through A3_HIP_LIB; the product library is never touched (tuning aid).
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
LIB=$ROOT/build/tuning/libaruco3_hip.so
for rep in 1 2; do
for cfg in ${CFGS:-"2 5 64 1" "2 5 64 0" "2 5 256 1" "2 5 256 0" "4 4 64 1"}; do
  set -- $cfg
  make -C $ROOT/aruco3_amd/csrc tuning TUNE_FLAGS="-DA3_D_KU=$1 -DA3_D_WAVES=$2 -DA3_D_THREADS=$3 -DA3_D_BLOCKED=${4:-1}" > /dev/null 2>&1 || exit 1
  (cd $ROOT && A3_HIP_LIB=$LIB python3 tools/kernel_probe.py 2>/dev/null | grep -E "decode +dbg=  0|decode +dbg= -1|decode +dbg= -2" | tr '\n' ' '; echo " <- kU=$1 waves=$2 threads=$3 blocked=${4:-1}")
done; done
