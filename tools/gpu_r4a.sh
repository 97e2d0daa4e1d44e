#!/bin/bash
# round 4, first GPU call: conditioning + sweep-mode tests after the ADVICE fixes, then the dense-level prologue traces
mkdir -p gpurun_out/r4a
timeout 1500 python -m pytest tests/test_gpu_sweep_modes.py tests/test_gpu_conditioning.py tests/test_gpu_parity.py -x -q > gpurun_out/r4a/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r4a/tests.log
tail -4 gpurun_out/r4a/tests.log
HEADN=12 bash tools/gpu_trace_env.sh dense0 PYDEM_SWEEP_DENSE=0
HEADN=12 bash tools/gpu_trace_env.sh dense3 PYDEM_SWEEP_DENSE=3
HEADN=14 bash tools/gpu_trace_env.sh dense5 PYDEM_SWEEP_DENSE=5
HEADN=16 bash tools/gpu_trace_env.sh dense8 PYDEM_SWEEP_DENSE=8
for d in 0 3 5; do PYDEM_SWEEP_DENSE=$d timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense $d', d['ms_per_step'], d['stages_ms']['sweep_ms'])"; done
