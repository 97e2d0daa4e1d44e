#!/bin/bash
# config 5 with the per-round trace of the pit drain paths (PYDEM_PATHS_DEBUG) and the conditioning debug lines
mkdir -p gpurun_out/c5
PYDEM_PATHS_DEBUG=1 timeout 300 python bench.py --config 5 --steps 1 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 > gpurun_out/c5/line.json 2> gpurun_out/c5/debug.txt
grep -c "round" gpurun_out/c5/debug.txt; grep "round" gpurun_out/c5/debug.txt | tail -30 | cut -c1-220
