cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2i
timeout 900 python -m pytest tests/test_gpu_edge_update.py tests/test_gpu_process_manager.py -m gpu -q -x > gpurun_out/r2i/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2i/gpu_tests.log
for b in 1024 256 128; do
PYDEM_EINC_BLOCK=$b PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r2i/pm_pool_16384_b$b.log 2>&1
done
