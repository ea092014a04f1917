#!/bin/bash
# Round 5: everything profiles/r05_* is made from, in three GPU calls (each well under 20 minutes):
#   bash tools/refresh_r05.sh a    bench (the driver's command) + kernel stats (isolated / overlapped) + rotation timeline
#   bash tools/refresh_r05.sh b    PMC: K1 traffic + issue counters, per-kernel HBM bytes of the chain, the noise workloads' kernel tables
#   bash tools/refresh_r05.sh c    microbenchmarks, stepping A/B, two-rank rehearsals (gloo), RCCL branch with one rank, gather verification, queue probe
# then: python tools/install_profiles.py r05
ROOT=$(cd "$(dirname "$0")/.." && pwd); cd "$ROOT"; OUT=$ROOT/gpurun_out/refresh; mkdir -p "$OUT"; export TMPDIR=/tmp
case ${1:-a} in
a)
  bash tools/refresh_profiles.sh bench stats
  OUTDIR=r05 bash tools/trace_rotation.sh > "$OUT/rotation.txt" 2>&1; tail -25 "$OUT/rotation.txt" ;;
b)
  bash tools/refresh_profiles.sh pmc
  bash tools/pmc_chain.sh > "$OUT/pmc_chain.log" 2>&1; cp gpurun_out/pmc_chain/summary.txt "$OUT/pmc_chain.txt"; cp gpurun_out/pmc_chain/pmc_chain.json "$OUT/pmc_chain.json"; tail -22 "$OUT/pmc_chain.txt"
  bash tools/noise_prof.sh > "$OUT/noise_prof.txt" 2>&1; cp gpurun_out/noise_c0/noise_kernel_stats.csv "$OUT/noise_c0_kernel_stats.csv"; cp gpurun_out/noise_c4/noise_kernel_stats.csv "$OUT/noise_c4_kernel_stats.csv"
  grep "^c0\|^c4" "$OUT/noise_prof.txt" ;;
c)
  make -C tools/micro > /dev/null 2>&1
  ./tools/micro/readbench k1 > "$OUT/readbench.txt" 2>&1; cat "$OUT/readbench.txt"
  ./tools/micro/scatterbench > "$OUT/scatterbench.txt" 2>&1; cat "$OUT/scatterbench.txt"
  bash tools/refresh_profiles.sh step dist
  python3 tools/queue_probe.py nccl 2>&1 | grep -v "^\[W\|amdgpu.ids" > "$OUT/queue_probe_nccl16.txt"
  HSA_ENABLE_IPC_MODE_LEGACY=0 python3 bench.py --frames 8 --steps 24 --warmup 4 --repeats 2 --isolated-launches 2 --device-synth --no-other-workloads --no-cpu-baseline --gpus 1 --force-dist --backend nccl --verify-gathers --gather-delay-us 4000 > "$OUT/gather_guard_on.log" 2>/dev/null
  HSA_ENABLE_IPC_MODE_LEGACY=0 python3 bench.py --frames 8 --steps 24 --warmup 4 --repeats 2 --isolated-launches 2 --device-synth --no-other-workloads --no-cpu-baseline --gpus 1 --force-dist --backend nccl --verify-gathers --gather-delay-us 4000 --no-gather-backpressure > "$OUT/gather_guard_off.log" 2>/dev/null
  python3 - "$OUT" <<'PY'
import json, sys
for f in ("gather_guard_on", "gather_guard_off"):
    d = json.loads([l for l in open(f"{sys.argv[1]}/{f}.log") if l.startswith("{")][-1]); g = d["gathered"]
    print(f, "collectives", g["verified_collectives"], "with wrong records", g["collectives_with_wrong_records"], "hw_queues", d["dist"]["hw_queues"])
PY
  ;;
esac
