timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r6/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6/tests.log; tail -4 gpurun_out/r6/tests.log
PYDEM_COND_BUILD=check timeout 1800 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py tests/test_gpu_large_configs.py -x -q -m gpu -k "not 8192" > gpurun_out/r6/pmtests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6/pmtests.log; tail -2 gpurun_out/r6/pmtests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6/smoke.log 2>&1; tail -1 gpurun_out/r6/smoke.log | cut -c1-120
SOAK_POOL=1 timeout 400 python tools/soak_pm.py 150 700000 > gpurun_out/r6/soak_pool.log 2>&1; grep "pm soak\|MISMATCH\|FAILED\|Error" gpurun_out/r6/soak_pool.log | tail -3
SOAK_POOL=1 SOAK_RCCL=1 timeout 300 python tools/soak_pm.py 90 710000 > gpurun_out/r6/soak_pool_rccl.log 2>&1; grep "pm soak\|MISMATCH\|FAILED\|Error" gpurun_out/r6/soak_pool_rccl.log | tail -3
SOAK_POOL=1 SOAK_SCALE=5 timeout 300 python tools/soak_pm.py 90 720000 > gpurun_out/r6/soak_pool_big.log 2>&1; grep "pm soak\|MISMATCH\|FAILED\|Error" gpurun_out/r6/soak_pool_big.log | tail -3
SOAK_POOL=1 SOAK_SCALE=5 PYDEM_COND_BUILD=check timeout 300 python tools/soak_pm.py 90 730000 > gpurun_out/r6/soak_pool_check.log 2>&1; grep "pm soak\|MISMATCH\|FAILED\|Error" gpurun_out/r6/soak_pool_check.log | tail -3
timeout 300 python tools/soak_pm.py 60 740000 > gpurun_out/r6/soak_serial.log 2>&1; grep "pm soak\|MISMATCH\|FAILED\|Error" gpurun_out/r6/soak_serial.log | tail -3
