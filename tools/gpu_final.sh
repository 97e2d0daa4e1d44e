#!/bin/bash
# end of round: the driver sequence (GPU tests, smoke, bench), the profile set, pool soaks through the queued waves
bash tools/gpu_full.sh
bash tools/collect_profiles.sh > gpurun_out/prof_collect.log 2>&1
tail -12 gpurun_out/prof_collect.log | cut -c1-300
mkdir -p gpurun_out/queue
S=${SOAK_SECONDS:-60}
SOAK_POOL=1 timeout $((S+120)) python tools/soak_pm.py $S 80000 > gpurun_out/queue/soak_pool.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" gpurun_out/queue/soak_pool.log
SOAK_POOL=1 SOAK_RCCL=1 timeout $((S+120)) python tools/soak_pm.py $S 90000 > gpurun_out/queue/soak_pool_rccl.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" gpurun_out/queue/soak_pool_rccl.log
for eb in 64 32; do
PYDEM_EVAL_BLOCKS=$eb PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 2>&1 | grep "edge fix-up wave" | tail -1 | sed "s/^/eval blocks $eb: /"
done
