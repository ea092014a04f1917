#!/bin/bash
# On the GPU box: per-kernel times of the per-dart sweeps for several grid caps (A3_*_BLOCKS).
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/sweep; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"; python3 bench.py --frames-cache /tmp/c2frames --no-cpu-baseline > "$OUT/b.log" 2>&1 || exit 1
cd /tmp
for cap in 2048 4096 8192 16384 32768; do
  export A3_LINK_BLOCKS=$cap A3_FIN_BLOCKS=$cap A3_SCATTER_BLOCKS=$cap
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o s$cap -- python3 "$ROOT/bench.py" --frames-cache /tmp/c2frames --no-cpu-baseline > "$OUT/s$cap.log" 2>&1
  python3 - "$OUT/s${cap}_kernel_stats.csv" $cap <<'PY'
import csv, sys
r = {x['Name'].split('(')[0]: float(x['AverageNs']) / 1e3 for x in csv.DictReader(open(sys.argv[1]))}
print(sys.argv[2], {k.replace('a3::', ''): round(v, 1) for k, v in r.items() if any(t in k for t in ('link', 'finalize', 'scatter'))})
PY
done
