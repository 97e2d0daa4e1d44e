#!/bin/bash
# queued waves: pool tests, pool-mode soaks (in-process strips / RCCL with one rank / plain launches), 8-tile timing
mkdir -p gpurun_out/queue
timeout 900 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py tests/test_gpu_large_configs.py -x -q -m gpu -k "not 8192" > gpurun_out/queue/tests.log 2>&1; tail -4 gpurun_out/queue/tests.log
S=${SOAK_SECONDS:-120}
SOAK_POOL=1 timeout $((S+120)) python tools/soak_pm.py $S 20000 > gpurun_out/queue/soak_pool.log 2>&1; tail -1 gpurun_out/queue/soak_pool.log
SOAK_POOL=1 SOAK_RCCL=1 timeout $((S+120)) python tools/soak_pm.py $((S/2)) 30000 > gpurun_out/queue/soak_pool_rccl.log 2>&1; tail -1 gpurun_out/queue/soak_pool_rccl.log
SOAK_POOL=1 PYDEM_EDGE_GRAPH=0 SOAK_SCALE=4 timeout $((S+120)) python tools/soak_pm.py $((S/2)) 40000 > gpurun_out/queue/soak_pool_nograph.log 2>&1; tail -1 gpurun_out/queue/soak_pool_nograph.log
for q in 16 8; do
PYDEM_EDGE_DEBUG=$([ $q = 8 ] && echo 1) PYDEM_EDGE_PROFILE=1 PYDEM_EDGE_QUEUE=$q PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/queue/pm_pool_q$q.log 2>&1
grep -v "per wave\|condensed edge round:" gpurun_out/queue/pm_pool_q$q.log | tail -4 | cut -c1-400
done
grep "condensed edge rounds:" gpurun_out/queue/pm_pool_q8.log | head -3
