#!/bin/bash
# pit drain paths: B against A (same-box), parity and soak on B
O=gpurun_out/r4t; mkdir -p $O
cp pydem_amd/lib/libpydem_hip.so.B pydem_amd/lib/libpydem_hip.so
timeout 900 python -m pytest tests/test_gpu_conditioning.py -x -q 2>&1 | tail -2
timeout 300 python tools/soak_conditioning_device.py 150 760000 > $O/soak.log 2>&1; tail -1 $O/soak.log
SOAK_NAN=1 timeout 200 python tools/soak_conditioning_device.py 60 40000 > $O/soak_nan.log 2>&1; tail -1 $O/soak_nan.log
SOAK_BIG=1 timeout 200 python tools/soak_conditioning_device.py 90 80000 > $O/soak_big.log 2>&1; tail -1 $O/soak_big.log
VARIANTS="A B" KEEP=B BENCH_ARGS="--config 5" bash tools/gpu_ab_lib.sh 2>&1 | tee $O/ab.log
PYDEM_PATHS_DEBUG=1 timeout 300 python bench.py --config 5 --steps 1 --warmup 1 --cpu-sample 0 2>&1 >/dev/null | grep -E "round [0-9]+:|pit drain" | tail -27 | cut -c1-200
