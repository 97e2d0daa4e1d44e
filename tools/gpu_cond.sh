cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
timeout 1500 python -m pytest tests/test_gpu_conditioning.py -m gpu -q -x > gpurun_out/r2g/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2g/gpu_tests.log
PYDEM_COND_DEBUG=1 PYDEM_PATHS_DEBUG=1 timeout 1200 python tools/run_config5.py 8192 > gpurun_out/r2g/cfg5_8192.log 2>&1
