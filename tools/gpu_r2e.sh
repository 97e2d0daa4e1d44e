cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2e
PYDEM_BOARD_CHECK=1 timeout 900 python -m pytest tests/test_gpu_edge_update.py tests/test_gpu_process_manager.py tests/test_gpu_soak.py -m gpu -q -x > gpurun_out/r2e/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2e/gpu_tests.log
PYDEM_EINC_COMPACT_MAX=0 timeout 900 python -m pytest tests/test_gpu_edge_update.py tests/test_gpu_process_manager.py -m gpu -q -x > gpurun_out/r2e/gpu_tests_noncompact.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2e/gpu_tests_noncompact.log
PM_WORKERS=8 PM_EDGE_MODE=pool PYDEM_EDGE_DEBUG=1 timeout 600 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r2e/pm_pool_16384.log 2>&1
PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 python tools/pm_multitile_timing.py 8192 8 > gpurun_out/r2e/pm_pool_8192.log 2>&1
