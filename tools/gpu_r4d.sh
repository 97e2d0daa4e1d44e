#!/bin/bash
mkdir -p gpurun_out/r4d
PYDEM_EDGE_DEBUG=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 python tools/pm_multitile_timing.py 16384 2 2>&1 | grep -E "condensed edge rounds:|n=16384" | head -8
timeout 2400 python -m pytest tests/test_gpu_edge_update.py tests/test_gpu_process_manager.py tests/test_gpu_soak.py tests/test_gpu_sweep_modes.py -x -q > gpurun_out/r4d/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4d/tests.log
tail -4 gpurun_out/r4d/tests.log
timeout 1500 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config4" > gpurun_out/r4d/large4.log 2>&1; tail -3 gpurun_out/r4d/large4.log
SOAK_POOL=1 timeout 300 python tools/soak_pm.py 150 0 > gpurun_out/r4d/soak_pool.log 2>&1; tail -1 gpurun_out/r4d/soak_pool.log
PYDEM_EDGE_SYNC=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4d/pm_pool_cond.log 2>&1; tail -4 gpurun_out/r4d/pm_pool_cond.log | cut -c1-700
PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4d/pm_pool_cond_async.log 2>&1; tail -1 gpurun_out/r4d/pm_pool_cond_async.log | cut -c1-300
