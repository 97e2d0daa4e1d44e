#!/bin/bash
mkdir -p gpurun_out/r4f
timeout 2400 python -m pytest tests/test_gpu_edge_update.py tests/test_gpu_process_manager.py tests/test_gpu_soak.py -x -q > gpurun_out/r4f/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4f/tests.log
tail -3 gpurun_out/r4f/tests.log
SOAK_POOL=1 timeout 300 python tools/soak_pm.py 150 0 > gpurun_out/r4f/soak_pool.log 2>&1; tail -1 gpurun_out/r4f/soak_pool.log
PYDEM_EDGE_SYNC=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4f/pm_pool_cond.log 2>&1; tail -4 gpurun_out/r4f/pm_pool_cond.log | cut -c1-900
PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4f/pm_pool_cond_async.log 2>&1; tail -1 gpurun_out/r4f/pm_pool_cond_async.log | cut -c1-300
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r4f/all.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4f/all.log; tail -4 gpurun_out/r4f/all.log
