mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_conditioning.py tests/test_gpu_large_configs.py tests/test_gpu_parity.py -m gpu -x -q -k "not config4" 2>&1 </dev/null | tail -5 > gpurun_out/cond_tests.txt
SOAK_NAN=1 timeout 200 python tools/soak_conditioning_device.py 90 790000 2>&1 </dev/null | tail -2 > gpurun_out/soak_cond.txt
timeout 200 python tools/soak_conditioning_device.py 90 800000 2>&1 </dev/null | tail -2 >> gpurun_out/soak_cond.txt
