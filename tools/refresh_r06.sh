#!/bin/bash
# Round 6: the one door to profiles/r06_*.  On the GPU box, three calls (each well under 20 minutes), then on the build host
# `python tools/install_profiles.py r06` copies the summaries from gpurun_out/refresh into profiles/:
#   bash tools/refresh_r06.sh a    the driver's bench command; rocprofv3 --kernel-trace --stats of the ISOLATED stepping (--no-pipeline: one
#                                  context, synchronous, every kernel alone -- what the roofline is computed from) and of the default,
#                                  overlapped stepping; the rotation timeline
#   bash tools/refresh_r06.sh b    PMC: K1 traffic + issue counters, per-kernel HBM bytes of the chain, the noise workloads' kernel tables
#   bash tools/refresh_r06.sh c    microbenchmarks, stepping A/B, rank rehearsals through the front door (gloo), the RCCL branch with one
#                                  rank, gather verification with and without the guard, the queue probe
# Every bench run leaves its compact line in <name>.json.log and everything else in <name>.detail.json (A3_BENCH_DETAIL).
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; OUT=$ROOT/gpurun_out/refresh; mkdir -p "$OUT" "$ROOT/gpurun_out/pmc"; export TMPDIR=/tmp
bench() {   # bench <name> <timeout> <args...>: stdout line -> $OUT/<name>.json.log, detail -> $OUT/<name>.detail.json
  local name=$1 to=$2; shift 2
  A3_BENCH_DETAIL=$OUT/$name.detail.json timeout -k 10 "$to" python3 bench.py "$@" > "$OUT/$name.json.log" 2> "$OUT/$name.err" || { echo "$name FAILED"; tail -5 "$OUT/$name.err"; return 1; }
  python3 tools/show_bench.py "$OUT/$name.json.log" | head -${SHOW:-3}
}
profiled() {   # profiled <name> <args...>: the same under rocprofv3 --kernel-trace --stats (the program itself after `--`, from /tmp)
  local name=$1; shift
  (cd /tmp && A3_BENCH_DETAIL=$OUT/$name.detail.json timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT" -o "$name" -- python3 "$ROOT/bench.py" "$@" > "$OUT/$name.log" 2>&1) || { tail -5 "$OUT/$name.log"; return 2; }
  grep '^{' "$OUT/$name.log" | tail -1 > "$OUT/$name.json.log"; python3 tools/show_bench.py "$OUT/$name.json.log" | head -2
}
case ${1:-a} in
a)
  SHOW=40 bench bench 900 --gpus 1 --steps 20 --warmup 5 || exit 1
  wc -c "$OUT/bench.json.log"
  profiled isolated --gpus 1 --steps 20 --warmup 5 --no-pipeline --device-synth --min-timed-s 1.0 --no-cpu-baseline --no-other-workloads
  profiled overlapped --gpus 1 --steps 20 --warmup 5 --device-synth --min-timed-s 1.0 --no-cpu-baseline --no-other-workloads
  OUTDIR=r06 bash tools/trace_rotation.sh > "$OUT/rotation.txt" 2>&1; tail -25 "$OUT/rotation.txt" ;;
b)
  rm -rf "$ROOT/gpurun_out/pmc"; mkdir -p "$ROOT/gpurun_out/pmc"
  bash tools/pmc_k1.sh > "$OUT/pmc.log" 2>&1 || { tail -5 "$OUT/pmc.log"; exit 3; }
  bash tools/pmc_issue.sh > "$OUT/pmc_issue.txt" 2>&1; head -14 "$OUT/pmc_issue.txt"
  bash tools/pmc_chain.sh > "$OUT/pmc_chain.log" 2>&1; cp gpurun_out/pmc_chain/summary.txt "$OUT/pmc_chain.txt"; cp gpurun_out/pmc_chain/pmc_chain.json "$OUT/pmc_chain.json"; tail -22 "$OUT/pmc_chain.txt"
  bash tools/noise_prof.sh > "$OUT/noise_prof.txt" 2>&1; cp gpurun_out/noise_c0/noise_kernel_stats.csv "$OUT/noise_c0_kernel_stats.csv"; cp gpurun_out/noise_c4/noise_kernel_stats.csv "$OUT/noise_c4_kernel_stats.csv"
  grep "^c0\|^c4" "$OUT/noise_prof.txt" ;;
c)
  make -C tools/micro > /dev/null 2>&1
  ./tools/micro/readbench k1 > "$OUT/readbench.txt" 2>&1; cat "$OUT/readbench.txt"
  ./tools/micro/scatterbench > "$OUT/scatterbench.txt" 2>&1; cat "$OUT/scatterbench.txt"
  timeout -k 10 400 python3 tools/ab_streams.py 256 48 5 shared:2:2:-1,own:2:2:-1,own:4:2:-1,own:4:2:-1:-1,own:4:2:-1:-1:0,own:4:2:2:-1,own:3:2:-1,own:4:2:-1:-1:1:same=1 2>&1 | grep -v amdgpu.ids > "$OUT/ab_streams.txt"; cut -c1-120 "$OUT/ab_streams.txt"
  bench rehearsal_n2_gloo 400 --gpus 2 --steps 20 --warmup 5 --backend gloo --device-synth --no-cpu-baseline
  bench rehearsal_n5_gloo 400 --gpus 5 --steps 10 --warmup 3 --frames 64 --backend gloo --device-synth --no-cpu-baseline --no-other-workloads
  bench rehearsal_c5_n2_gloo 400 --workload c5 --gpus 2 --steps 20 --warmup 5 --backend gloo --device-synth --no-cpu-baseline
  bench bench_c5 300 --workload c5 --device-synth --no-other-workloads
  HSA_ENABLE_IPC_MODE_LEGACY=0 bench force_dist_nccl_1rank 300 --gpus 1 --force-dist --backend nccl --device-synth --no-cpu-baseline --no-other-workloads
  python3 tools/queue_probe.py nccl 2>&1 | grep -v "^\[W\|amdgpu.ids" > "$OUT/queue_probe_nccl16.txt"
  G="--frames 8 --steps 24 --warmup 4 --repeats 2 --isolated-launches 2 --device-synth --no-other-workloads --no-cpu-baseline --gpus 1 --force-dist --backend nccl --verify-gathers --gather-delay-us 4000"
  HSA_ENABLE_IPC_MODE_LEGACY=0 SHOW=0 bench gather_guard_on 300 $G
  HSA_ENABLE_IPC_MODE_LEGACY=0 SHOW=0 bench gather_guard_off 300 $G --no-gather-backpressure
  python3 - "$OUT" <<'PY'
import json, sys
for f in ("gather_guard_on", "gather_guard_off"):
    d = json.load(open(f"{sys.argv[1]}/{f}.detail.json")); g = d["gathered"]
    print(f, "collectives", g["verified_collectives"], "with wrong records", g["collectives_with_wrong_records"], "hw_queues", d["dist"]["hw_queues"])
PY
  ;;
esac
ls "$OUT" | tr '\n' ' '
