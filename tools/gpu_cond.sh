cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
PYDEM_PATHS_DEBUG=1 timeout 1500 python -m pytest tests/test_gpu_conditioning.py -m gpu -q -x -k "pit_paths" > gpurun_out/r2g/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2g/gpu_tests.log
