#!/bin/bash
# pit drain paths: speculation window / large simulations per round
mkdir -p gpurun_out/coop
for cfg in ${PW_CFGS:-"131072 2048" "131072 4096" "131072 8192" "196608 4096" "131072 2048" "131072 4096"}; do
set -- $cfg
PYDEM_PATHS_DEBUG=1 PYDEM_PATHS_WINDOW=$1 PYDEM_PATHS_BIG=$2 timeout 300 python bench.py --config 5 --steps 2 --warmup 1 > gpurun_out/coop/b.json 2> gpurun_out/coop/b.err
python - "$cfg" <<'PY'
import json,sys
d=json.load(open('gpurun_out/coop/b.json'))
print(sys.argv[1], round(d['ms_per_step'],1), round(d['stages_ms']['pit_paths_ms'],1))
PY
grep "pit drain" gpurun_out/coop/b.err | tail -1
done
