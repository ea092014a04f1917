#!/bin/bash
# On the GPU box: K1 time for several rows-per-wave settings (A3_ROWS_PER_WAVE), interleaved twice to average out drift.
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"
python3 bench.py --frames-cache /tmp/c2frames --no-cpu-baseline > /dev/null 2>&1 || exit 1
for rep in 1 2; do
for rows in ${ROWS:-61 76 91 106 121 136 166 196}; do
  r=$(A3_ROWS_PER_WAVE=$rows python3 bench.py --steps 10 --warmup 2 --frames-cache /tmp/c2frames --no-cpu-baseline 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['stage_ms_per_step']['threshold'], d['value'])")
  echo "rows=$rows -> threshold_ms fps: $r"
done; done
