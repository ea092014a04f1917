#!/bin/bash
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"; OUT=$ROOT/gpurun_out/r4; mkdir -p $OUT
timeout -k 10 500 python3 bench.py > $OUT/bench_default.json 2> $OUT/bench_default.err || { tail -20 $OUT/bench_default.err; exit 1; }
python3 tools/show_bench.py $OUT/bench_default.json
timeout -k 10 300 python3 bench.py --workload c5 --device-synth --no-other-workloads > $OUT/bench_c5.json 2> $OUT/bench_c5.err || { tail -20 $OUT/bench_c5.err; exit 2; }
python3 tools/show_bench.py $OUT/bench_c5.json
timeout -k 10 400 python3 bench.py --workload c5 --gpus 2 --backend gloo --device-synth --no-other-workloads --no-cpu-baseline --steps 4 --repeats 3 > $OUT/bench_c5_n2.json 2> $OUT/bench_c5_n2.err || { tail -20 $OUT/bench_c5_n2.err; exit 3; }
python3 tools/show_bench.py $OUT/bench_c5_n2.json
