#!/bin/bash
# block tier of the pit search: 256 (A) / 1024 (B) / 512 (C) threads per pit on config 5; pit parity on the winner candidates
O=gpurun_out/r4v; mkdir -p $O
for v in B C; do cp pydem_amd/lib/libpydem_hip.so.$v pydem_amd/lib/libpydem_hip.so; timeout 900 python -m pytest tests/test_gpu_pits.py tests/test_gpu_large_configs.py -x -q -k "pit or config5 or config3" 2>&1 | tail -2; done
VARIANTS="A B C" KEEP=A BENCH_ARGS="--config 5" bash tools/gpu_ab_lib.sh 2>&1 | cut -c1-190 | tee $O/ab.log
