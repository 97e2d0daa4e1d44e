#!/bin/bash
# tile passes in phases (B) against one launch per pass (A): parity, same-box bench, threshold of the phased regime
O=gpurun_out/r4x; mkdir -p $O
cp pydem_amd/lib/libpydem_hip.so.B pydem_amd/lib/libpydem_hip.so
timeout 1500 python -m pytest tests/test_gpu_sweep_modes.py tests/test_gpu_parity.py tests/test_gpu_large_configs.py tests/test_gpu_edge_update.py -x -q > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
VARIANTS="A B C" KEEP=B bash tools/gpu_ab_lib.sh 2>&1 | cut -c1-150 | tee $O/ab.log
for M in 0 4096 65536 1000000000; do
  PYDEM_SWEEP_PHASES_MIN=$M timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('PHASES_MIN=$M', round(d['ms_per_step'],2), 'sweep', round(d['stages_ms']['sweep_ms'],2))"
done
PYDEM_SWEEP_DEBUG=1 timeout 300 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>&1 >/dev/null | grep -E "tile passes|listed tile pass" | head -14 | cut -c1-140
