cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2g
PYDEM_PATHS_DEBUG=1 timeout 900 python tools/run_config5.py 2048 > gpurun_out/r2g/cfg5_2048.log 2>&1
PYDEM_PATHS_DEBUG=1 timeout 900 python tools/run_config5.py 4096 > gpurun_out/r2g/cfg5_4096.log 2>&1
PYDEM_PATHS_DEBUG=1 timeout 1200 python tools/run_config5.py 8192 > gpurun_out/r2g/cfg5_8192.log 2>&1
