mkdir -p gpurun_out
KEEP=B bash tools/gpu_ab_lib.sh > gpurun_out/ab_pits.txt 2>&1
PYDEM_PITS_HANDOVER=1 timeout 300 python tools/leak_probe.py 10 2>&1 | tail -6 > gpurun_out/leak_ab.txt
timeout 900 python -m pytest tests/test_gpu_pits.py tests/test_gpu_soak.py -m gpu -x -q 2>&1 | tail -4 > gpurun_out/pits_tests.txt
