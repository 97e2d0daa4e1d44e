mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_sweep_modes.py tests/test_gpu_large_configs.py -m gpu -x -q -k "conditioning_schedules or config5" 2>&1 </dev/null | tail -5 > gpurun_out/cond_tests.txt
timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null </dev/null > gpurun_out/c5.json
