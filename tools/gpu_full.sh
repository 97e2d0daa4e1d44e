cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2f
timeout 1200 python -m pytest tests -m gpu -q -x > gpurun_out/r2f/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2f/gpu_tests.log
python bench.py --steps 5 --warmup 2 > gpurun_out/r2f/bench.json 2> gpurun_out/r2f/bench.err
PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r2f/pm_pool_16384.log 2>&1
