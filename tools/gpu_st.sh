#!/bin/bash
# stencil iteration: parity tests that exercise slopes/directions, then the stencil timing of the bench
mkdir -p gpurun_out/st
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py -x -q > gpurun_out/st/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/st/tests.log
tail -5 gpurun_out/st/tests.log
timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --roof-iters 10 > gpurun_out/st/bench.json 2> gpurun_out/st/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/st/bench.json'))
print(d['roofline']); print(d['ms_per_step'], d['stages_ms'])
PY
timeout 900 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config3" > gpurun_out/st/large.log 2>&1; tail -3 gpurun_out/st/large.log
