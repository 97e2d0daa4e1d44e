#!/bin/bash
# round 4: condensed edge rounds -- parity tests, pool soak, 8-tile timing against the cell-by-cell form; stencil occupancy A/B
mkdir -p gpurun_out/r4c
timeout 1800 python -m pytest tests/test_gpu_edge_update.py tests/test_gpu_process_manager.py tests/test_gpu_soak.py -x -q > gpurun_out/r4c/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4c/tests.log
tail -15 gpurun_out/r4c/tests.log
timeout 900 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config4" > gpurun_out/r4c/large4.log 2>&1; tail -5 gpurun_out/r4c/large4.log
SOAK_POOL=1 timeout 400 python tools/soak_pm.py 240 0 > gpurun_out/r4c/soak_pool.log 2>&1; tail -3 gpurun_out/r4c/soak_pool.log
PYDEM_EDGE_SYNC=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4c/pm_pool_cond.log 2>&1; tail -4 gpurun_out/r4c/pm_pool_cond.log | cut -c1-400
PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4c/pm_pool_cond_async.log 2>&1; tail -1 gpurun_out/r4c/pm_pool_cond_async.log | cut -c1-300
PYDEM_EDGE_COND=0 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r4c/pm_pool_cell.log 2>&1; tail -2 gpurun_out/r4c/pm_pool_cell.log | cut -c1-300
for o in 1 4; do PYDEM_STENCIL_OCC=$o timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 20 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('occ $o', d['ms_per_step'], d['roofline_stencil']['avg_kernel_ms'], d['roofline_stencil']['back_to_back_ms'])"; done
