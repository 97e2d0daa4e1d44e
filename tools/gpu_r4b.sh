#!/bin/bash
# dense-level prologue, second look: per-level cost after the counter fix, totals for 3..12 levels
mkdir -p gpurun_out/r4b
for d in 5 8 12; do HEADN=$((d+3)) bash tools/gpu_trace_env.sh dense$d PYDEM_SWEEP_DENSE=$d; done
for d in 0 3 5 6 8 10 12 16; do PYDEM_SWEEP_DENSE=$d timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('dense $d', d['ms_per_step'], d['stages_ms']['sweep_ms'])"; done
