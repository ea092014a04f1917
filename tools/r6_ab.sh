#!/bin/bash
# On the GPU box: library builds A/B'd on ONE box (boxes of the pool differ by more than most effects).
#   bash tools/r6_ab.sh <tag> "<lib> <lib> ..." [workload c2|noise|c4|one] [frames] [headline rounds, 0 = none]
# per library: stage times of isolated batches (events), the per-kernel table of the same run under rocprofv3 --kernel-trace --stats;
# then the headline stepping (tools/ab_streams.py, four contexts free-running) in alternating processes.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"
TAG=$1; LIBS=$2; WL=${3:-c2}; FR=${4:-256}; HR=${5:-2}
OUT=$ROOT/gpurun_out/$TAG; mkdir -p "$OUT"; export TMPDIR=/tmp
for lib in $LIBS; do
  name=$(echo "$lib" | tr '/.' '__')
  A3_HIP_LIB=$ROOT/$lib timeout -k 10 300 python3 tools/r6_iso.py $FR 12 $WL 2>/dev/null | tail -1 | tee -a "$OUT/iso.txt" | cut -c1-230
  (cd /tmp && A3_HIP_LIB=$ROOT/$lib timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$name" -o k -- python3 "$ROOT/tools/r6_iso.py" $FR 12 $WL > "$OUT/$name.log" 2>&1)
  f=$(find "$OUT/$name" -name "k_kernel_stats.csv" | head -1)
  python3 - "$f" "$lib" <<'PY' | tee -a "$OUT/kernels.txt"
import csv, sys
print("==", sys.argv[2])
tot = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    if "a3::" in r["Name"] and float(r["Percentage"]) > 0.3:
        tot += float(r["AverageNs"]) / 1e3 * (int(r["Calls"]) / 15.0 if False else 1)
        print(f"  {r['Name'][:56]:58s} calls {r['Calls']:>5s} avg_us {float(r['AverageNs']) / 1e3:8.1f} {float(r['Percentage']):5.1f}%")
PY
done
if [ "$HR" != "0" ]; then
  for r in $(seq $HR); do
    for lib in $LIBS; do
      echo -n "$lib  "; A3_HIP_LIB=$ROOT/$lib timeout -k 10 300 python3 tools/ab_streams.py 256 40 3 own:4:2:-1 2>/dev/null | tail -1 | cut -c1-150
    done
  done | tee -a "$OUT/headline.txt"
fi
