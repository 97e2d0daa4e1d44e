#!/bin/bash
# small-window path simulations: A = one kernel, B/C/D = growth + path kernels with the trail in global memory at 5 / 6 / 8 wavefronts per SIMD
O=gpurun_out/r4t; mkdir -p $O
for v in C D; do cp pydem_amd/lib/libpydem_hip.so.$v pydem_amd/lib/libpydem_hip.so; timeout 600 python -m pytest tests/test_gpu_conditioning.py -x -q 2>&1 | tail -1; done
cp pydem_amd/lib/libpydem_hip.so.C pydem_amd/lib/libpydem_hip.so
timeout 300 python tools/soak_conditioning_device.py 100 770000 2>&1 | tail -1
VARIANTS="A B C D" KEEP=A BENCH_ARGS="--config 5" bash tools/gpu_ab_lib.sh 2>&1 | cut -c1-40,95-175
