#!/bin/bash
# fill_flats: exactness against the host twin, then timings per setting (cap, workgroups, one XCD only, lists at or below `min` stay with the single workgroup)
mkdir -p gpurun_out/coop
timeout 1200 python -m pytest tests/test_gpu_conditioning.py -x -q > gpurun_out/coop/tests.log 2>&1; tail -2 gpurun_out/coop/tests.log
for cfg in "0 64 0 0" "8192 32 0 0" "32768 32 0 0" "32768 64 0 0" "8192 32 0 1024" "8192 32 0 4096" "131072 64 0 0" "0 64 0 0" "8192 32 0 0"; do
set -- $cfg
PYDEM_FLAT_COOP=$1 PYDEM_FLAT_COOP_WG=$2 PYDEM_FLAT_COOP_XCD=$3 PYDEM_FLAT_COOP_MIN=$4 timeout 300 python bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/coop/b.json 2> gpurun_out/coop/b.err
python - "$cfg" <<'PY'
import json,sys
d=json.load(open('gpurun_out/coop/b.json'))
print(sys.argv[1], round(d['ms_per_step'],1), round(d['stages_ms']['fill_flats_ms'],1))
PY
done
