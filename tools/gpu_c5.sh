#!/bin/bash
mkdir -p gpurun_out/c5
timeout 1500 python -m pytest tests/test_gpu_conditioning.py tests/test_gpu_large_configs.py -x -q > gpurun_out/c5/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/c5/tests.log
tail -4 gpurun_out/c5/tests.log
PYDEM_PATHS_DEBUG=1 PYDEM_COND_DEBUG=1 timeout 600 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/c5/bench5.json 2> gpurun_out/c5/bench5.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/c5/bench5.json'))
print(d['value'], d['ms_per_step'], d.get('stages_ms'))
PY
timeout 900 python tools/run_config5.py 8192 2>&1 | tail -8
