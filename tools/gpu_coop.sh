#!/bin/bash
# fill_flats: timings per setting of the resident sweeps (cap, workgroups, one XCD only, lists at or below `min` stay with the single workgroup)
mkdir -p gpurun_out/coop
for cfg in "8192 32 0 0" "4096 32 0 0" "6144 32 0 0" "12288 32 0 0" "8192 16 0 0" "8192 48 0 0" "8192 64 0 0" "8192 32 0 512" "8192 32 0 0"; do
set -- $cfg
PYDEM_FLAT_COOP=$1 PYDEM_FLAT_COOP_WG=$2 PYDEM_FLAT_COOP_XCD=$3 PYDEM_FLAT_COOP_MIN=$4 timeout 300 python bench.py --config 5 --steps 3 --warmup 1 > gpurun_out/coop/b.json 2> gpurun_out/coop/b.err
python - "$cfg" <<'PY'
import json,sys
d=json.load(open('gpurun_out/coop/b.json'))
print(sys.argv[1], round(d['ms_per_step'],1), round(d['stages_ms']['fill_flats_ms'],1))
PY
done
