#!/bin/bash
# fill_flats with several sweeps per pass (k_flat_batch): parity tests, soak against the host twin, config 5 with and without it
O=gpurun_out/fb; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_conditioning.py -x -q > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
timeout 400 python tools/soak_conditioning_device.py ${SOAK_S:-200} > $O/soak.log 2>&1; tail -2 $O/soak.log
SOAK_NAN=1 timeout 300 python tools/soak_conditioning_device.py ${SOAK_S2:-120} 40000 > $O/soak_nan.log 2>&1; tail -2 $O/soak_nan.log
SOAK_BIG=1 timeout 300 python tools/soak_conditioning_device.py ${SOAK_S2:-120} 80000 > $O/soak_big.log 2>&1; tail -2 $O/soak_big.log
for V in 16384 0; do
  PYDEM_FLAT_BATCH=$V PYDEM_COND_DEBUG=1 timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench5_$V.json 2> $O/bench5_$V.err
  python - <<PY
import json
d=json.loads(open('$O/bench5_$V.json').read().strip().splitlines()[-1])
print('PYDEM_FLAT_BATCH=$V', d['ms_per_step'], {k: round(v, 2) for k, v in d['stages_ms'].items() if k in ('fill_flats_ms', 'pit_paths_ms', 'terrain_ms')})
PY
  grep "fill_flats:" $O/bench5_$V.err | tail -2
done
