#!/bin/bash
# On the GPU box: the HBM-side bytes of EVERY kernel of a batch, settled per kernel (FETCH_SIZE must be doubled only
# where the requests are 128-byte ones).  gfx950 has byte-accurate fabric counters -- TCC_EA0_RDREQ_DRAM_32B (a 64-byte request counts 2, a
# 128-byte one 4) and TCC_EA0_WRREQ_WRITE_DRAM_32B / _ATOMIC_DRAM_32B -- collected here beside the request counters FETCH_SIZE / WRITE_SIZE are
# derived from (TCC_EA0_RDREQ, _32B, TCC_BUBBLE = 128-byte requests; TCC_EA0_WRREQ, _64B).  Four passes of at most four TCC counters, counters
# only, on the isolated stepping (one synchronous batch at a time: every kernel alone).  Output: gpurun_out/pmc_chain/summary.txt + .json
ROOT=$(cd "$(dirname "$0")/.." && pwd); OUT=$ROOT/gpurun_out/pmc_chain; rm -rf "$OUT"; mkdir -p "$OUT"; export TMPDIR=/tmp
ARGS="$ROOT/bench.py --device-synth --no-cpu-baseline --no-other-workloads --no-pipeline --repeats 1 --steps 3 --warmup 1 --isolated-launches 2 ${BENCH_ARGS:-}"
cd /tmp
PASSES=(
  "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_32B_sum TCC_BUBBLE_sum"
  "TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_WRITE_DRAM_32B_sum TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum"
  "FETCH_SIZE GRBM_GUI_ACTIVE"
  "WRITE_SIZE TCC_HIT_sum TCC_MISS_sum"
)
i=0
for P in "${PASSES[@]}"; do
  i=$((i+1))
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $P --output-format csv -d "$OUT" -o p$i -- python3 $ARGS > "$OUT/p$i.log" 2>&1 || { echo "pass $i ($P) FAILED"; tail -5 "$OUT/p$i.log"; exit 1; }
done
python3 - "$OUT" > "$OUT/summary.txt" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
dur = collections.defaultdict(list)
for path in glob.glob(out + "/*_kernel_trace.csv"):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "a3::" in k: dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(out + "/*_counter_collection.csv"):
    per = collections.defaultdict(float); names = {}
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0]
        if "a3::" not in k: continue
        per[(r["Dispatch_Id"], r["Counter_Name"])] += float(r["Counter_Value"]); names[r["Dispatch_Id"]] = k
    for (d, c), v in per.items(): acc[names[d]][c].append(v)
rows, js = [], {}
for k in acc:
    v = {c: sum(x) / len(x) for c, x in acc[k].items()}
    us = sum(dur[k]) / max(len(dur[k]), 1) / 1e3
    g = lambda n: v.get(n, float("nan"))
    rd_exact = g("TCC_EA0_RDREQ_DRAM_32B_sum") * 32 / 1e6
    wr_exact = (g("TCC_EA0_WRREQ_WRITE_DRAM_32B_sum") + g("TCC_EA0_WRREQ_ATOMIC_DRAM_32B_sum")) * 32 / 1e6
    fetch = g("FETCH_SIZE") * 1024 / 1e6; write = g("WRITE_SIZE") * 1024 / 1e6
    rows.append((rd_exact + wr_exact, k, us, g("TCC_EA0_RDREQ_sum") / 1e6, g("TCC_EA0_RDREQ_32B_sum") / 1e6, g("TCC_BUBBLE_sum") / 1e6, rd_exact, fetch, g("TCC_EA0_WRREQ_sum") / 1e6,
                 g("TCC_EA0_WRREQ_64B_sum") / 1e6, wr_exact, write))
    js[k] = dict(v, us_under_pmc=us, read_MB_exact=rd_exact, write_MB_exact=wr_exact)
print("per launch, mean over dispatches; 'exact' = TCC_EA0_*_DRAM_32B x 32 B; FETCH_SIZE / WRITE_SIZE as rocprofv3 derives them (MB = 1e6 B)")
print(f"{'kernel':44s} {'us(pmc)':>8s} | {'RDREQ M':>8s} {'32B M':>7s} {'128B M':>7s} {'read MB exact':>13s} {'FETCH_SIZE MB':>13s} {'exact/FETCH':>11s} | {'WRREQ M':>8s} {'64B M':>7s} {'write MB exact':>14s} {'WRITE_SIZE MB':>13s}")
tot = [0.0, 0.0, 0.0, 0.0]
for r in sorted(rows, reverse=True):
    _, k, us, rq, r32, bub, rde, fe, wq, w64, wre, wr = r
    print(f"{k.replace('a3::','').replace('void ','')[:44]:44s} {us:8.1f} | {rq:8.2f} {r32:7.2f} {bub:7.2f} {rde:13.1f} {fe:13.1f} {rde / max(fe, 1e-9):11.2f} | {wq:8.2f} {w64:7.2f} {wre:14.1f} {wr:13.1f}")
    if "k_grey_threshold" not in k and "k_synth" not in k and "k_pack" not in k:
        tot[0] += rde; tot[1] += wre; tot[2] += fe; tot[3] += us
print(f"{'the chain (everything but the threshold kernel)':44s} {tot[3]:8.1f} | read exact {tot[0]:.1f} MB (FETCH_SIZE {tot[2]:.1f}), written {tot[1]:.1f} MB")
json.dump(js, open(out + "/pmc_chain.json", "w"), indent=1)
PY
cat "$OUT/summary.txt"
