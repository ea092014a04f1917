#!/bin/bash
# On the GPU box: two builds of the library A/B'd in alternating processes on ONE box (boxes of the pool differ by more than most effects).
#   tools/ab_libs.sh <lib A> <lib B> [rounds] [ab_streams spec]
cd "$(dirname "$0")/.."
A=$1; B=$2; R=${3:-3}; SPEC=${4:-own:4:2:-1}
for r in $(seq $R); do
  for lib in $A $B; do
    echo -n "$lib  "; A3_HIP_LIB=$PWD/$lib python tools/ab_streams.py 256 40 3 $SPEC 2>/dev/null | tail -1 | cut -c1-150
  done
done
