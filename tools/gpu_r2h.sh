cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2h
timeout 1500 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_large_configs.py -m gpu -q -x > gpurun_out/r2h/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2h/gpu_tests.log
