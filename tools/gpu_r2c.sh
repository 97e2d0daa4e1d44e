cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2c
PM_WORKERS=8 PM_EDGE_MODE=pool PYDEM_EDGE_DEBUG=1 timeout 600 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/r2c/pm_pool_16384_prof.log 2>&1
