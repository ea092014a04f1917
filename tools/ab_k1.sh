#!/bin/bash
# A/B the threshold kernel variants under tools/k1_variants on ONE box, interleaved (run on the GPU box).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd $ROOT/aruco3_amd/csrc
cp k_threshold.hip /tmp/k_threshold.keep
python $ROOT/bench.py --steps 2 --warmup 1 --frames-cache /tmp/c2frames --no-cpu-baseline > /dev/null 2>&1
for round in 1 2 3; do
  for v in ${A3_VARIANTS:-/tmp/k1_variants}/*.hip; do
    cp $v k_threshold.hip
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++20 -fPIC -ffp-contract=off -c k_threshold.hip -o k_threshold.o || exit 1
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libaruco3_hip.so a3_api.o k_threshold.o k_contours.o k_decode.o || exit 1
    r=$(cd $ROOT && python bench.py --steps 10 --warmup 3 --frames-cache /tmp/c2frames --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stage_ms_per_step']['threshold'], d['value'])")
    echo "round $round $(basename $v): $r"
  done
done
cp /tmp/k_threshold.keep k_threshold.hip
