set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2a
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r2a/gpu_tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r2a/gpu_tests.log
tail -5 gpurun_out/r2a/gpu_tests.log
for sz in 8192 16384; do
  timeout 600 python tools/pm_multitile_timing.py $sz 8 > gpurun_out/r2a/pm_serial_$sz.log 2>&1
  PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 python tools/pm_multitile_timing.py $sz 8 > gpurun_out/r2a/pm_pool_$sz.log 2>&1
  tail -4 gpurun_out/r2a/pm_serial_$sz.log gpurun_out/r2a/pm_pool_$sz.log
done
PM_WORKERS=8 PM_EDGE_MODE=pool PYDEM_EDGE_DEBUG=1 timeout 600 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r2a/pm_pool_16384_dbg.log 2>&1
