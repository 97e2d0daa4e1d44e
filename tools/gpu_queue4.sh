#!/bin/bash
# after a change to the operator build: pool / edge tests, a short pool soak, build breakdown at 8 x 16384^2
mkdir -p gpurun_out/queue
timeout 900 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py tests/test_gpu_large_configs.py -x -q -m gpu -k "not 8192" > gpurun_out/queue/tests.log 2>&1; grep -n "passed\|failed" gpurun_out/queue/tests.log | tail -2
S=${SOAK_SECONDS:-45}
SOAK_POOL=1 SOAK_SCALE=6 timeout $((S+120)) python tools/soak_pm.py $S 70000 > gpurun_out/queue/soak_pool_big.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" gpurun_out/queue/soak_pool_big.log
for thr in 1 4; do
PYDEM_COND_THREADS=$thr PYDEM_EDGE_DEBUG=1 PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/queue/pm_pool_thr$thr.log 2>&1
echo "threads $thr"; grep "condensed edge rounds: [0-9]" gpurun_out/queue/pm_pool_thr$thr.log | tail -3 | cut -c1-330; grep "edge fix-up wave\|^n=" gpurun_out/queue/pm_pool_thr$thr.log | tail -2 | cut -c1-300
done
