#!/bin/bash
# On the GPU box: rebuild k_decode with other (samples in flight per lane, waves per SIMD) and time it (tuning aid).
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd $ROOT/aruco3_amd/csrc
for rep in 1 2; do
for cfg in ${CFGS:-"2 5 64 1" "2 5 64 0" "2 5 256 1" "2 5 256 0" "4 4 64 1"}; do
  set -- $cfg
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt -DA3_D_KU=$1 -DA3_D_WAVES=$2 -DA3_D_THREADS=$3 -DA3_D_BLOCKED=${4:-1} -c k_decode.hip -o k_decode.o || exit 1
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaruco3_hip.so a3_api.o k_threshold.o k_contours.o k_decode.o k_synth.o || exit 1
  (cd $ROOT && python3 tools/kernel_probe.py 2>/dev/null | grep -E "decode +dbg=  0|decode +dbg= -1|decode +dbg= -2" | tr '\n' ' '; echo " <- kU=$1 waves=$2 threads=$3 blocked=${4:-1}")
done; done
