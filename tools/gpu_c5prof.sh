#!/bin/bash
# config 5 with a -DPYDEM_PATHS_PROF build of cond_paths.hip (pydem_amd/lib/libpydem_hip.so.P, linked beforehand): ticks per phase of
# the pit-path simulations by window class, next to the per-round trace
L=pydem_amd/lib/libpydem_hip.so; mkdir -p gpurun_out/c5; cp $L $L.keep; cp $L.P $L
PYDEM_PATHS_DEBUG=1 timeout 300 python bench.py --config 5 --steps 1 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 > gpurun_out/c5/prof_line.json 2> gpurun_out/c5/prof_debug.txt
cp $L.keep $L
grep -v "round [0-9]*:" gpurun_out/c5/prof_debug.txt | tail -8 | cut -c1-600; grep -A2 "round [0-9]*:" gpurun_out/c5/prof_debug.txt | tail -72 | cut -c1-260
