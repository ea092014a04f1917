#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh'): the driver's bench command (its host-rendered frames kept in a
# cache file so that the profiled runs read the very same frames and nothing forks under the profiler), the same command under
# rocprofv3 --kernel-trace --stats, the PMC passes (tools/pmc_k1.sh), the streaming-read microbenchmark and the 2-rank gloo
# rehearsal; everything lands in gpurun_out/refresh/.  tools/install_profiles.py then copies the summaries into profiles/.
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/refresh
rm -rf "$OUT" "$ROOT/gpurun_out/pmc"; mkdir -p "$OUT" "$ROOT/gpurun_out/pmc"; export TMPDIR=/tmp
cd "$ROOT"
timeout -k 10 600 python3 bench.py --gpus 1 --steps 20 --warmup 5 --frames-cache /tmp/c2frames > "$OUT/bench.json.log" 2> "$OUT/bench.err" || exit 1
cut -c1-600 "$OUT/bench.json.log"
cd /tmp
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o stats -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --frames-cache /tmp/c2frames --no-cpu-baseline --no-other-workloads > "$OUT/stats.log" 2>&1 || exit 2
tail -c 400 "$OUT/stats.log"; echo
bash "$ROOT/tools/pmc_k1.sh" > "$OUT/pmc.log" 2>&1 || exit 3
cd "$ROOT"
[ -x tools/micro/readbench ] && timeout -k 5 120 ./tools/micro/readbench > "$OUT/readbench.txt" 2>&1
[ -x tools/micro/scatterbench ] && timeout -k 5 120 ./tools/micro/scatterbench > "$OUT/scatterbench.txt" 2>&1
# the N = 2 line through the bench's own front door (no launcher in the command): two child ranks on the one leased GPU, gloo
timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --device-synth --no-cpu-baseline > "$OUT/rehearsal_n2_gloo.log" 2> "$OUT/rehearsal_n2_gloo.err" || echo "rehearsal failed"
timeout -k 10 300 python3 tools/ab_overlap.py 256 40 5 2 > "$OUT/ab_overlap.txt" 2>&1
bash "$ROOT/tools/pmc_issue.sh" > "$OUT/pmc_issue.txt" 2>&1
grep '^{' "$OUT/rehearsal_n2_gloo.log" | cut -c1-300
ls "$OUT" "$ROOT/gpurun_out/pmc"
