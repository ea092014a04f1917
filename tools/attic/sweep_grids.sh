#!/bin/bash
# On the GPU box: per-kernel times for several grid sizes of the contour-stage kernels (A3_* knobs).
# (the A3_* grid knobs are only read by a -DA3_TUNING build: build/tuning/, loaded through A3_HIP_LIB)
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/sweep; mkdir -p "$OUT"; export TMPDIR=/tmp
make -C "$ROOT/aruco3_amd/csrc" tuning > /dev/null 2>&1 || exit 1
export A3_HIP_LIB=$ROOT/build/tuning/libaruco3_hip.so
cd /tmp
ARGS="$ROOT/bench.py --device-synth --no-cpu-baseline --no-other-workloads --repeats 2"
for cfg in "1024 512" "2048 512" "3072 1024" "4096 2048" "512 512"; do
  set -- $cfg
  export A3_QUAD_BLOCKS64=$1 A3_QUAD_BLOCKS16=$2
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o q$1 -- python3 $ARGS > "$OUT/q$1.log" 2>&1
  python3 - "$OUT/q$1_kernel_stats.csv" "$cfg" <<'PY'
import csv, sys
r = {x['Name'].split('(')[0]: float(x['AverageNs']) / 1e3 for x in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], {k.replace('a3::', ''): round(v, 1) for k, v in r.items() if any(t in k for t in ('quads',))})
PY
done
