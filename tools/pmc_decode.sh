#!/bin/bash
# On the GPU box: is the decode stage's sampling bound by the memory system's REQUEST rate?  (counters instead of
# argument.)  Three things side by side:
#   1. tools/micro/scatterbench: the kernel's tap-read pattern with everything else taken away, at the kernel's occupancy
#   2. k_decode itself, truncated after the sampling (a3_debug_kernel_time, tools/kernel_probe.py: dbg -2) and whole
#   3. PMC passes (counters only, each pass on its own) for k_decode in the synchronous bench and for the microbenchmark:
#      L1 -> L2 read requests, L2 requests, L2 -> fabric read requests, L2 hits / misses, L1 request latency and stall cycles
# Output: gpurun_out/pmc_decode/summary.txt
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/pmc_decode; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
cd "$ROOT"
make -C tools/micro scatterbench > /dev/null 2>&1
{
echo "== scatterbench (tools/micro/scatterbench.hip) =="; timeout -k 5 120 ./tools/micro/scatterbench
echo "== k_decode through a3_debug_kernel_time (dbg 0: projection + decode, -1: decode alone, -2: stop after sampling, -3: after Otsu, -4: after bits) =="
timeout -k 5 200 python3 tools/kernel_probe.py 2>/dev/null | grep -E "^decode|^stats"
} > "$OUT/summary.txt" 2>&1
ARGS="$ROOT/bench.py --device-synth --no-cpu-baseline --no-other-workloads --no-pipeline --repeats 1 --steps 3 --warmup 1"
cd /tmp
# Counter passes: each pass asks a hardware block for no more counters than it holds (round 3's second pass asked the L1 (TCP) for
# five and aborted: "Request exceeds the capabilities of the hardware to collect" -- at most two TCP counters per pass now).  A pass
# that fails ends the script with a non-zero exit code: a summary must never be written from a run with a hole in it.
PASSES=(
  "TCP_TCC_READ_REQ_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum GRBM_GUI_ACTIVE"
  "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_LATENCY_sum"
  "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum"
  "TCP_GATE_EN1_sum TCP_TCC_READ_REQ_sum"
  "FETCH_SIZE TCC_EA0_RDREQ_32B_sum TCC_TAG_STALL_sum SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_INSTS_VMEM_RD"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT" -o dec$i -- python3 $ARGS > "$OUT/dec$i.log" 2>&1 || { echo "pass $i (bench: $P) FAILED"; tail -5 "$OUT/dec$i.log"; exit 1; }
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT" -o mic$i -- "$ROOT/tools/micro/scatterbench" > "$OUT/mic$i.log" 2>&1 || { echo "pass $i (micro: $P) FAILED"; tail -5 "$OUT/mic$i.log"; exit 1; }
done
python3 - "$OUT" >> "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
dur = collections.defaultdict(list)      # kernel -> durations (ns) from the kernel traces
for path in glob.glob(out + "/*_kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_decode" in k or "k_scatter<" in k:
            dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/*_counter_collection.csv"):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "k_decode" in k or "k_scatter<" in k:
            per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = k
    for (d, c), v in per.items(): acc[names[d]][c].append(v)
print("== PMC per launch (mean over dispatches; under the counters a launch takes longer than alone: rates use the traced duration of the same passes) ==")
for k in sorted(acc):
    us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    v = {c: sum(x) / len(x) for c, x in acc[k].items()}
    def g(n, scale=1.0, fmt="{:.2f}"):      # a counter that was not collected prints as n/a, never as 0
        return fmt.format(v[n] / scale) if n in v else "n/a"
    def ratio(a, b, fmt="{:.2f}"):
        return fmt.format(v[a] / max(v[b], 1.0)) if a in v and b in v else "n/a"
    def rate(n):
        return f"{v[n] / max(us, 1e-9):8.0f} /us" if n in v else "n/a"
    print(f"{k[:70]}\n   duration under PMC {us:8.1f} us   L1->L2 read req {g('TCP_TCC_READ_REQ_sum', 1e6)} M ({rate('TCP_TCC_READ_REQ_sum')})"
          f"   L2 req {g('TCC_REQ_sum', 1e6)} M ({rate('TCC_REQ_sum')})   L2->fabric rd {g('TCC_EA0_RDREQ_sum', 1e6)} M ({rate('TCC_EA0_RDREQ_sum')}; 32 B: {g('TCC_EA0_RDREQ_32B_sum', 1e6)} M)"
          f"\n   L2 hit/miss {g('TCC_HIT_sum', 1e6)}/{g('TCC_MISS_sum', 1e6)} M   FETCH_SIZE {g('FETCH_SIZE', 1024, '{:.1f}')} MiB   L1 accesses {g('TCP_TOTAL_CACHE_ACCESSES_sum', 1e6)} M"
          f"   L1 read latency / req {ratio('TCP_TCC_READ_REQ_LATENCY_sum', 'TCP_TCC_READ_REQ_sum', '{:.0f}')} clk   L1 pending-stall cyc {g('TCP_PENDING_STALL_CYCLES_sum', 1e6, '{:.1f}')} M, TCR->TCP stall {g('TCP_TCR_TCP_STALL_CYCLES_sum', 1e6, '{:.1f}')} M of gate-enabled {g('TCP_GATE_EN1_sum', 1e6, '{:.1f}')} M"
          f"   wave wait_any/cycle {ratio('SQ_WAIT_ANY', 'SQ_WAVE_CYCLES')}   vmem rd insts {g('SQ_INSTS_VMEM_RD', 1e6)} M")
PY
cat "$OUT/summary.txt"
