#!/bin/bash
mkdir -p gpurun_out/queue
timeout 900 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py -x -q -m gpu > gpurun_out/queue/tests.log 2>&1; grep -n "passed\|failed" gpurun_out/queue/tests.log | tail -2
C5_KEEPS="" bash tools/gpu_queue_prof.sh 2>&1 | grep "k_board_eval\|QTile\|k_sched\|pack_line_q\|edge fix-up wave"
PYDEM_EDGE_DEBUG=1 PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 2>&1 | grep "flush (interior" | tail -8
