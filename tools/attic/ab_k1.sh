#!/bin/bash
# On the GPU box: K1 of two builds of the library in alternation (A3_HIP_LIB selects the copy), same frames, same process shape:
# stage times and frames/s of the bench.  LIBS="path1 path2 ..." (default: the product library and build/oldk1)
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
cd "$ROOT"
B="python3 bench.py --device-synth --no-cpu-baseline --no-other-workloads --repeats ${REPEATS:-40} --steps 20 --warmup 3"
for rep in $(seq 1 ${REPS:-3}); do
  for lib in ${LIBS:-aruco3_amd/libaruco3_hip.so build/oldk1/libaruco3_hip.so}; do
    A3_HIP_LIB=$ROOT/$lib $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', 'k1_ms', d['roofline']['avg_launch_ms'], 'frac', d['roofline']['frac'], 'stages', d['stage_ms_per_step'], 'fps', d['value'], d['frames_with_all_ids_correct'])"
  done
done
