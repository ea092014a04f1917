#!/bin/bash
# On the GPU box: threshold-stage parity tests, then K1 time for the output-burst settings (device-rendered frames, no CPU legs).
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"
OUT=$ROOT/gpurun_out/r2k1; mkdir -p "$OUT"
timeout -k 10 600 python3 -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "threshold or baseline_configs or bgra or strided or other_threshold or noise_frames" > "$OUT/k1_tests.log" 2>&1
echo "k1 tests rc=$? $(tail -1 $OUT/k1_tests.log)"
[ "$(grep -c failed $OUT/k1_tests.log)" != "0" ] && { tail -40 "$OUT/k1_tests.log"; exit 1; }
B="python3 bench.py --device-synth --no-cpu-baseline --no-other-workloads --repeats 5 --steps 20 --warmup 3"
for rep in 1 2 3; do
for fl in ${FLUSH:-0 64 128 240}; do
  r=$(A3_K1_FLUSH=$fl $B 2>/dev/null | tail -1)
  echo "$r" | python3 -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('flush=$fl', 'k1_ms', d['stage_ms_per_step']['threshold'], 'frac', d['roofline']['frac'], 'fps', d['value'], 'ids', d['frames_with_all_ids_correct'])"
done; done 2>&1 | tee "$OUT/sweep.log"
./tools/micro/readbench | sed "s/|/\n/g" | grep -E "read-only|short x64|burst at" | head -3
