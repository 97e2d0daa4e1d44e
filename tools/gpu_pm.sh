#!/bin/bash
mkdir -p gpurun_out/pm
timeout 1500 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_parity.py -x -q > gpurun_out/pm/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/pm/tests.log
tail -25 gpurun_out/pm/tests.log
