#!/bin/bash
# resident visits (K5e): sweep tests, then the bench with the switch at several list sizes (same box)
mkdir -p gpurun_out/res
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep_modes.py tests/test_gpu_pits.py -x -q > gpurun_out/res/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/res/tests.log
tail -4 gpurun_out/res/tests.log
for sw in 0 1024 4096 16384 65536 0 4096; do
PYDEM_SWEEP_RESIDENT=$sw timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 5 > gpurun_out/res/bench_$sw.json 2> gpurun_out/res/bench_$sw.err
python - $sw <<'PY'
import json,sys
d=json.load(open('gpurun_out/res/bench_%s.json'%sys.argv[1]))
print(sys.argv[1], d['ms_per_step'], d['stages_ms'].get('sweep'), d['sweep'])
PY
done
