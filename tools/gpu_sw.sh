#!/bin/bash
# sweep iteration: the tests that exercise the sweep, then the bench
mkdir -p gpurun_out/sw
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep_modes.py tests/test_gpu_soak.py tests/test_gpu_pits.py -x -q > gpurun_out/sw/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/sw/tests.log
tail -4 gpurun_out/sw/tests.log
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 5 > gpurun_out/sw/bench.json 2> gpurun_out/sw/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/sw/bench.json'))
print(d['ms_per_step'], d['stages_ms'], d['sweep'])
PY
timeout 900 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config3" > gpurun_out/sw/large.log 2>&1; tail -3 gpurun_out/sw/large.log
