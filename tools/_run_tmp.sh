PYDEM_COND_BUILD=check timeout 1800 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py tests/test_gpu_large_configs.py -x -q -m gpu -k "not 8192" > gpurun_out/r6/pmtests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6/pmtests.log; tail -2 gpurun_out/r6/pmtests.log
TAG=_dev STEPS=fixup TAILN=3 bash tools/gpu_r6.sh; grep "device build" gpurun_out/r6/fixup_dev.log | tail -8 | cut -c150-360
PYDEM_CB_WIDE=0 TAG=_dev_wide0 STEPS=fixup TAILN=1 bash tools/gpu_r6.sh; grep "device build" gpurun_out/r6/fixup_dev_wide0.log | tail -8 | cut -c150-360
C5_KEEPS="" bash tools/gpu_queue_prof.sh 2>&1 | grep "k_cb_\|can\|ort" | cut -c1-200
PYDEM_CB_WIDE=0 C5_KEEPS="" bash tools/gpu_queue_prof.sh 2>&1 | grep "k_cb_" | cut -c1-200
