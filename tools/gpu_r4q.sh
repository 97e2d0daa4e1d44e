#!/bin/bash
# A: round-4 addressing; B: round loop of the tile visits with scalar bases + neighbour table; C: B with eight listed wavefronts per SIMD.
# fill_flats with queued passes.
O=gpurun_out/r4q; mkdir -p $O
cp pydem_amd/lib/libpydem_hip.so.B pydem_amd/lib/libpydem_hip.so
timeout 1500 python -m pytest tests/test_gpu_sweep_modes.py tests/test_gpu_parity.py tests/test_gpu_conditioning.py -x -q > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -3 $O/tests.log
VARIANTS="A B C" KEEP=B bash tools/gpu_ab_lib.sh 2>&1 | tee $O/ab.log
timeout 300 python tools/soak_conditioning_device.py 100 > $O/soak.log 2>&1; tail -1 $O/soak.log
SOAK_BIG=1 timeout 200 python tools/soak_conditioning_device.py 80 80000 > $O/soak_big.log 2>&1; tail -1 $O/soak_big.log
for V in 16384 65536 0; do
  PYDEM_FLAT_BATCH=$V PYDEM_COND_DEBUG=1 timeout 600 python bench.py --config 5 --steps 3 --warmup 1 --cpu-sample 0 > $O/bench5_$V.json 2> $O/bench5_$V.err
  python - <<PY
import json
d=json.loads(open('$O/bench5_$V.json').read().strip().splitlines()[-1])
print('PYDEM_FLAT_BATCH=$V', d['ms_per_step'], {k: round(v, 2) for k, v in d['stages_ms'].items() if k in ('fill_flats_ms', 'pit_paths_ms', 'terrain_ms')})
PY
  grep "fill_flats:" $O/bench5_$V.err | tail -2
done
