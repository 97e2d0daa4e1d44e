cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2d
PYDEM_BOARD_CHECK=1 timeout 900 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_soak.py -m gpu -q -x > gpurun_out/r2d/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2d/gpu_tests.log
PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 python tools/pm_multitile_timing.py 16384 8 prof > gpurun_out/r2d/pm_pool_16384_prof.log 2>&1
