#!/bin/bash
# On the GPU box: the bench under each of a list of environment settings, interleaved REPS times so that drift of the box shows.
# The knobs are only read by a -DA3_TUNING build: this script makes one (build/tuning/, `make tuning`) and loads it through
# A3_HIP_LIB; the product library ignores the environment.  ENVS="A=1;A=2 B=3;..."  (an empty entry = defaults)
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp
make -C "$ROOT/aruco3_amd/csrc" tuning > /dev/null 2>&1 || exit 1
export A3_HIP_LIB=$ROOT/build/tuning/libaruco3_hip.so
cd "$ROOT"
B="python3 bench.py --device-synth --no-cpu-baseline --no-other-workloads --repeats ${REPEATS:-7} --steps 20 --warmup 3"
IFS=';' read -ra LIST <<< "${ENVS:-;}"
for rep in $(seq 1 ${REPS:-2}); do
  for e in "${LIST[@]}"; do
    env $e $B 2>/dev/null | tail -1 | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('[$e]', 'stages', d['stage_ms_per_step'], 'fps', d['value'], d['frames_with_all_ids_correct'])"
  done
done
