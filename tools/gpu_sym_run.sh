SYM_TEST=100000000 SYM_LIST="0 4096 32768 0 4096 32768" SYM_TRACE="0 4096 32768" bash tools/gpu_sym.sh
PYDEM_SWEEP_SYM=40 timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q 2>&1 | tail -3
