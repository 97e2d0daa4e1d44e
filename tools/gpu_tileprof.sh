#!/bin/bash
# phase cycles of the tile passes (library built with -DPYDEM_TILE_PROF): all passes, passes >= 3, passes >= 11, >= 25
mkdir -p gpurun_out/sw
for first in 1 3 11 25; do
  echo "== passes >= $first"
  PYDEM_TILE_DEBUG=$((4 + 256 * first)) timeout 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 1 2>&1 >/dev/null | grep "tile phases"
done > gpurun_out/sw/tileprof.txt
cat gpurun_out/sw/tileprof.txt
