#!/bin/bash
# Run on the GPU box (gpurun -- 'bash tools/refresh_profiles.sh [part ...]'): everything profiles/<tag>_* is made from, into
# gpurun_out/refresh/ (tools/install_profiles.py copies the summaries into profiles/).  Parts (default: all, in this order):
#   bench      the driver's bench command (host-rendered frames kept in a cache file so that the profiled runs read the very same
#              frames and nothing forks under the profiler)
#   stats      rocprofv3 --kernel-trace --stats of (a) the ISOLATED stepping (--no-pipeline: one context, synchronous, every kernel
#              alone -- the durations the roofline is computed from) and (b) the default, overlapped stepping (durations in company)
#   pmc        the PMC passes of the isolated stepping (tools/pmc_k1.sh, pmc_issue.sh)
#   micro      readbench, scatterbench, valubench, k1_concurrency
#   step       tools/ab_streams.py (stepping arrangements in one process), tools/spin_probe.py
#   dist       the N = 2 lines through the bench's own front door (gloo, two ranks on the one GPU): config 2 and config 5
#   decode     tools/pmc_decode.sh
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=$ROOT/gpurun_out/refresh
mkdir -p "$OUT" "$ROOT/gpurun_out/pmc"; export TMPDIR=/tmp
PARTS=${*:-bench stats pmc micro step dist decode}
cd "$ROOT"
for part in $PARTS; do
  echo "== $part"
  case $part in
  bench)
    timeout -k 10 900 python3 bench.py --gpus 1 --steps 20 --warmup 5 > "$OUT/bench.json.log" 2> "$OUT/bench.err" || { tail -5 "$OUT/bench.err"; exit 1; }
    python3 tools/show_bench.py "$OUT/bench.json.log" ;;
  stats)
    cd /tmp
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o isolated -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --no-pipeline --device-synth --min-timed-s 1.0 --no-cpu-baseline --no-other-workloads > "$OUT/isolated.log" 2>&1 || { tail -5 "$OUT/isolated.log"; exit 2; }
    timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o overlapped -- python3 "$ROOT/bench.py" --gpus 1 --steps 20 --warmup 5 --device-synth --min-timed-s 1.0 --no-cpu-baseline --no-other-workloads > "$OUT/overlapped.log" 2>&1 || { tail -5 "$OUT/overlapped.log"; exit 2; }
    cd "$ROOT"
    for f in isolated overlapped; do grep '^{' "$OUT/$f.log" | tail -1 > "$OUT/$f.json.log"; python3 tools/show_bench.py "$OUT/$f.json.log" | head -2; done ;;
  pmc)
    rm -rf "$ROOT/gpurun_out/pmc"; mkdir -p "$ROOT/gpurun_out/pmc"
    bash tools/pmc_k1.sh > "$OUT/pmc.log" 2>&1 || { tail -5 "$OUT/pmc.log"; exit 3; }
    bash tools/pmc_issue.sh > "$OUT/pmc_issue.txt" 2>&1; head -14 "$OUT/pmc_issue.txt" ;;
  micro)
    for b in readbench scatterbench valubench; do [ -x tools/micro/$b ] && timeout -k 5 200 ./tools/micro/$b > "$OUT/$b.txt" 2>&1; done
    timeout -k 10 200 python3 tools/k1_concurrency.py 256 10 2>&1 | grep -v amdgpu.ids > "$OUT/k1_concurrency.txt"; cat "$OUT/k1_concurrency.txt" ;;
  step)
    timeout -k 10 400 python3 tools/ab_streams.py 256 48 5 shared:2:2:-1,own:2:2:-1,own:4:2:-1,own:4:2:-1:-1,own:4:2:-1:-1:0,own:4:2:2:-1,own:4:2:-1:4,own:3:2:-1:-1,own:5:2:-1:-1,own:4:2:-1:-1:1:same=1 2>&1 | grep -v amdgpu.ids > "$OUT/ab_streams.txt"; cut -c1-120 "$OUT/ab_streams.txt"
    timeout -k 10 600 python3 tools/spin_probe.py 256 40 2>&1 | grep -v amdgpu.ids > "$OUT/spin_probe.txt"; cut -c1-150 "$OUT/spin_probe.txt" ;;
  dist)
    timeout -k 10 400 python3 bench.py --gpus 2 --steps 20 --warmup 5 --backend gloo --device-synth --no-cpu-baseline > "$OUT/rehearsal_n2_gloo.log" 2> "$OUT/rehearsal_n2_gloo.err" || echo "c2 rehearsal failed"
    timeout -k 10 400 python3 bench.py --workload c5 --gpus 2 --steps 20 --warmup 5 --backend gloo --device-synth --no-cpu-baseline > "$OUT/rehearsal_c5_n2_gloo.log" 2> "$OUT/rehearsal_c5_n2_gloo.err" || echo "c5 rehearsal failed"
    timeout -k 10 300 python3 bench.py --workload c5 --device-synth --no-other-workloads > "$OUT/bench_c5.json.log" 2> "$OUT/bench_c5.err" || echo "c5 bench failed"
    timeout -k 10 300 python3 bench.py --gpus 1 --force-dist --backend nccl --device-synth --no-cpu-baseline --no-other-workloads > "$OUT/force_dist_nccl_1rank.log" 2> "$OUT/force_dist_nccl_1rank.err" || echo "nccl 1-rank failed"
    for f in rehearsal_n2_gloo rehearsal_c5_n2_gloo bench_c5.json force_dist_nccl_1rank; do python3 tools/show_bench.py "$OUT/$f.log" | head -3; done ;;
  decode)
    bash tools/pmc_decode.sh > "$OUT/pmc_decode.log" 2>&1 || { echo "pmc_decode FAILED"; tail -8 "$OUT/pmc_decode.log"; }
    cp "$ROOT/gpurun_out/pmc_decode/summary.txt" "$OUT/pmc_decode.txt" 2>/dev/null ;;
  esac
done
ls "$OUT"
