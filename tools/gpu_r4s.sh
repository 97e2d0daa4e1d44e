#!/bin/bash
# plane cache: the whole GPU suite (tiles reuse each other's planes), the drop-in call; phase timers of the medium-window pit simulations
O=gpurun_out/r4s; mkdir -p $O
timeout 2400 python -m pytest tests -m gpu -x -q > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
PYDEM_BENCH_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --roof-iters 0 > $O/bench.json 2> $O/bench.err; grep host_to_host $O/bench.err | cut -c1-330
PYDEM_PLANE_CACHE_GB=0 PYDEM_BENCH_DEBUG=1 timeout 600 python bench.py --cpu-sample 0 --roof-iters 0 > $O/bench_nocache.json 2> $O/bench_nocache.err; grep host_to_host $O/bench_nocache.err | cut -c1-330
timeout 300 python tools/time_host_to_host.py c5 2>&1 | cut -c1-300
cp pydem_amd/lib/libpydem_hip.so $O/keep.so; cp pydem_amd/lib/libpydem_hip.so.P pydem_amd/lib/libpydem_hip.so
PYDEM_PATHS_DEBUG=1 timeout 600 python bench.py --config 5 --steps 1 --warmup 0 --cpu-sample 0 > $O/bench5_prof.json 2> $O/bench5_prof.err; grep -E "large-window|pit drain paths" $O/bench5_prof.err | cut -c1-400
cp $O/keep.so pydem_amd/lib/libpydem_hip.so; rm $O/keep.so
