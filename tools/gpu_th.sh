#!/bin/bash
# tile height 32 / 64 of the sweep (library variants in tools/ab): sweep tests with the 64-row library, then the bench line per variant
mkdir -p gpurun_out/th
cp pydem_amd/lib/libpydem_hip.so /tmp/lib_orig.so
cp tools/ab/lib_th64.so pydem_amd/lib/libpydem_hip.so
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep_modes.py tests/test_gpu_pits.py tests/test_gpu_large_configs.py -x -q -k "not lds" > gpurun_out/th/tests64.log 2>&1; tail -3 gpurun_out/th/tests64.log
for f in tools/ab/lib_th32.so tools/ab/lib_th64.so tools/ab/lib_th32.so tools/ab/lib_th64.so; do
cp $f pydem_amd/lib/libpydem_hip.so
PYDEM_SWEEP_DEBUG=1 timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 3 > gpurun_out/th/b.json 2> gpurun_out/th/b.err
python - "$f" <<'PY'
import json,sys
d=json.load(open('gpurun_out/th/b.json'))
print(sys.argv[1], round(d['ms_per_step'],2), 'sweep', round(d['stages_ms']['sweep_ms'],2), 'passes', d['sweep'].get('kernel_launches'))
PY
done
cp /tmp/lib_orig.so pydem_amd/lib/libpydem_hip.so
