#!/bin/bash
# grouped sub-passes: sweep tests, bench for several sub-pass counts, per-pass trace
mkdir -p gpurun_out/sub
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep_modes.py tests/test_gpu_soak.py tests/test_gpu_pits.py tests/test_gpu_edge_update.py -x -q > gpurun_out/sub/tests.log 2>&1; grep -n "passed\|failed" gpurun_out/sub/tests.log | tail -3
for sp in 4 8 12; do
  PYDEM_SWEEP_VISITS=$sp timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 0 > gpurun_out/sub/b$sp.json 2> gpurun_out/sub/b$sp.err
  python - <<PY
import json
d=json.load(open('gpurun_out/sub/b$sp.json'))
print('subpasses $sp: ms_per_step %.2f sweep_ms %.2f launches %s' % (d['ms_per_step'], d['stages_ms']['sweep_ms'], d['sweep']['kernel_launches']))
PY
done
HEADN=12 bash tools/gpu_trace_env.sh v8 PYDEM_SWEEP_VISITS=8
timeout 900 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config3" > gpurun_out/sub/large.log 2>&1; grep -n "passed\|failed" gpurun_out/sub/large.log | tail -2
