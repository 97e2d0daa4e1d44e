#!/bin/bash
mkdir -p gpurun_out/r4i
timeout 2400 python -m pytest tests/test_gpu_conditioning.py tests/test_gpu_parity.py tests/test_gpu_process_manager.py -x -q > gpurun_out/r4i/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4i/tests.log; tail -3 gpurun_out/r4i/tests.log
timeout 1500 python -m pytest tests/test_gpu_large_configs.py tests/test_gpu_sweep_modes.py -x -q > gpurun_out/r4i/tests2.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4i/tests2.log; tail -3 gpurun_out/r4i/tests2.log
SOAK_NAN=1 timeout 400 python tools/soak_conditioning_device.py 200 7708 2>&1 | tail -1
timeout 300 python bench.py --config 5 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('config5', d['ms_per_step'], d['stages_ms'])"
