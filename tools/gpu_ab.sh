#!/bin/bash
# A/B of library variants in tools/ab/*.so on one box: one bench line per variant (AB_ARGS, default the config-5 line; the product
# library is restored at the end).  Variants: compile one unit with -D... and link it with the objects in pydem_amd/lib/.
mkdir -p gpurun_out/ab
cp pydem_amd/lib/libpydem_hip.so /tmp/lib_orig.so
for f in tools/ab/*.so; do
cp $f pydem_amd/lib/libpydem_hip.so
PYDEM_PATHS_DEBUG=1 timeout 300 python bench.py ${AB_ARGS:---config 5} --steps 3 --warmup 1 > gpurun_out/ab/b.json 2> gpurun_out/ab/b.err
python - "$f" <<'PY'
import json,sys
d=json.load(open('gpurun_out/ab/b.json'))
print(sys.argv[1], round(d['ms_per_step'],2), {k: round(v, 2) for k, v in d['stages_ms'].items() if k.endswith('_ms')})
PY
grep "pit drain" gpurun_out/ab/b.err | tail -1
done
cp /tmp/lib_orig.so pydem_amd/lib/libpydem_hip.so
