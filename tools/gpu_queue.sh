#!/bin/bash
# Edge fix-up (queued waves, operator build) on the GPU box -- ONE driver, the parts chosen by PARTS (space-separated):
#   tests         pool / edge tests (+ PYTEST_EXTRA, e.g. tests/test_gpu_large_configs.py -k "not 8192")
#   soak          pool-mode soak, in-process strips            (SOAK_SECONDS, SOAK_SEED, SOAK_SCALE)
#   soak_rccl     the same over RCCL with one rank             (SOAK_GRAPH=0: plain launches)
#   timing        tools/pm_multitile_timing.py ${QUEUE_N:-16384} 8 for every entry of VARIANTS; an entry is a list of
#                 VAR=value pairs joined by commas, e.g. VARIANTS="PYDEM_EDGE_QUEUE=0 PYDEM_EDGE_QUEUE=16,PYDEM_EDGE_GRAPH=0
#                 PYDEM_COND_BUILD=host PM_RCCL=1,PYDEM_EDGE_QUEUE=16 PM_IN_FLIGHT=auto" (default: one run, defaults)
#   prof          kernel trace of the fix-up's kernels (gpu_queue_prof.sh)
# Output under gpurun_out/queue/.  (Replaces gpu_queue.sh .. gpu_queue6.sh of round 5: the same runs, as parameters.)
O=gpurun_out/queue; mkdir -p $O
S=${SOAK_SECONDS:-60}
for part in ${PARTS:-tests timing}; do
case $part in
tests)     timeout 1500 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py ${PYTEST_EXTRA} -x -q -m gpu > $O/tests.log 2>&1; grep -n "passed\|failed" $O/tests.log | tail -2 ;;
soak)      SOAK_POOL=1 SOAK_SCALE=${SOAK_SCALE:-1} timeout $((S+120)) python tools/soak_pm.py $S ${SOAK_SEED:-140000} > $O/soak_pool.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" $O/soak_pool.log ;;
soak_rccl) SOAK_POOL=1 SOAK_RCCL=1 PYDEM_EDGE_GRAPH=${SOAK_GRAPH:-1} SOAK_SCALE=${SOAK_SCALE:-1} timeout $((S+120)) python tools/soak_pm.py $S ${SOAK_SEED:-150000} > $O/soak_pool_rccl.log 2>&1; grep "pm soak\|MISMATCH\|FAILED" $O/soak_pool_rccl.log ;;
timing)    for v in ${VARIANTS:-default}; do
             tag=$(echo $v | tr -c 'A-Za-z0-9=,\n' '_')
             ( [ $v != default ] && export $(echo $v | tr ',' ' '); PYDEM_EDGE_DEBUG=1 PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py ${QUEUE_N:-16384} 8 > $O/pm_pool_$tag.log 2>&1 )
             echo "== $v"; grep "condensed edge rounds[^:]*: [0-9]" $O/pm_pool_$tag.log | tail -2 | cut -c1-330; grep "edge fix-up wave\|^n=" $O/pm_pool_$tag.log | tail -2 | cut -c1-300
           done ;;
prof)      C5_KEEPS="" bash tools/gpu_queue_prof.sh ;;
*)         echo "unknown part $part" ;;
esac
done
