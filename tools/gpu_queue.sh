#!/bin/bash
# queued waves of the edge fix-up: the pool-mode tests, then the 8-tile timing with and without the queue
mkdir -p gpurun_out/queue
timeout 900 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py -x -q -m gpu > gpurun_out/queue/tests.log 2>&1; tail -15 gpurun_out/queue/tests.log
if [ -n "$QUEUE_TIMING" ]; then
for q in 0 16 16g0 64; do
PYDEM_EDGE_GRAPH=$([ $q = 16g0 ] && echo 0 || echo 1) PYDEM_EDGE_PROFILE=1 PYDEM_EDGE_QUEUE=${q%g0} PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py ${QUEUE_N:-16384} 8 > gpurun_out/queue/pm_pool_q$q.log 2>&1
grep -v "per wave" gpurun_out/queue/pm_pool_q$q.log | tail -6 | cut -c1-400
done
fi
