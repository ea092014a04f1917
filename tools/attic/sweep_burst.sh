#!/bin/bash
# On the GPU box: the grid-size knobs of the contour-stage kernels (a -DA3_TUNING build, A3_HIP_LIB) under the BURST stepping --
# four instances of every kernel run together there, so the sizes tuned for a kernel running alone need not be the best.
ROOT=$(cd "$(dirname "$0")/.." && pwd); export TMPDIR=/tmp; cd "$ROOT"
export A3_HIP_LIB=$ROOT/build/tuning/libaruco3_hip.so
run() { echo -n "$* : "; env "$@" timeout -k 10 200 python3 tools/ab_streams.py 256 48 ${ROUNDS:-4} own:4:2:0:-1:1 2>&1 | grep median | cut -c18-70; }
run A3_NONE=1
run A3_LINK_BLOCKS=2048
run A3_LINK_BLOCKS=8192
run A3_FIN_BLOCKS=768
run A3_FIN_BLOCKS=3072
run A3_SCATTER_BLOCKS=4096
run A3_SCATTER_BLOCKS=16384
run A3_QUAD_BLOCKS64=1280 A3_QUAD_BLOCKS16=2048
run A3_QUAD_BLOCKS64=5120 A3_QUAD_BLOCKS16=8192
run A3_SELECT_BLOCKS=512
run A3_NONE=2
