#!/bin/bash
# the queued collective (areas as doubles, masks as bytes): pool / edge tests, pool soaks over RCCL with one rank (graphs and plain launches) and in-process, 8-tile timing over RCCL
mkdir -p gpurun_out/queue
timeout 900 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py -x -q -m gpu > gpurun_out/queue/tests.log 2>&1; grep -n "passed\|failed" gpurun_out/queue/tests.log | tail -2
S=${SOAK_SECONDS:-60}
SOAK_POOL=1 SOAK_RCCL=1 timeout $((S+120)) python tools/soak_pm.py $S 110000 > gpurun_out/queue/soak_pool_rccl.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" gpurun_out/queue/soak_pool_rccl.log
SOAK_POOL=1 SOAK_RCCL=1 PYDEM_EDGE_GRAPH=0 SOAK_SCALE=4 timeout $((S+120)) python tools/soak_pm.py $((S/2)) 120000 > gpurun_out/queue/soak_pool_rccl_nograph.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" gpurun_out/queue/soak_pool_rccl_nograph.log
SOAK_POOL=1 timeout $((S+120)) python tools/soak_pm.py $((S/2)) 130000 > gpurun_out/queue/soak_pool.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" gpurun_out/queue/soak_pool.log
for q in 16 0; do
PM_RCCL=1 PYDEM_EDGE_QUEUE=$q PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 2>&1 | grep "edge fix-up wave\|^n=" | tail -2 | sed "s/^/rccl one rank, queue $q: /" | cut -c1-260
done
