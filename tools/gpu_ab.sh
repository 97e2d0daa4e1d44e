#!/bin/bash
# A/B of library variants in tools/ab/*.so on one box: config-5 bench line per variant (the product library is restored at the end)
mkdir -p gpurun_out/ab
cp pydem_amd/lib/libpydem_hip.so /tmp/lib_orig.so
for f in tools/ab/*.so; do
cp $f pydem_amd/lib/libpydem_hip.so
PYDEM_PATHS_DEBUG=1 timeout 300 python bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err
python - "$f" <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab/b.json'))
print(sys.argv[1], round(d['ms_per_step'],1), 'pit paths', round(d['stages_ms']['pit_paths_ms'],1), 'fill_flats', round(d['stages_ms']['fill_flats_ms'],1))
PY
grep "pit drain" gpurun_out/ab/b.err | tail -1
done
cp /tmp/lib_orig.so pydem_amd/lib/libpydem_hip.so
