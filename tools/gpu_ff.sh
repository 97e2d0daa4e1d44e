#!/bin/bash
# fill_flats of config 5 under the batch switches (PYDEM_FLAT_BATCH = list length at which the 16-sweep LDS passes take over,
# PYDEM_FLAT_BATCH_REGIONS = rows of their arrival table): stage time per variant, three alternating repetitions
mkdir -p gpurun_out/ff
for rep in 1 2 3; do
for v in ${VARIANTS:-16384:4096 32768:4096 65536:4096 131072:4096}; do
  PYDEM_FLAT_BATCH=${v%%:*} PYDEM_FLAT_BATCH_REGIONS=${v##*:} timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$v rep $rep: ms_per_step %.2f fill_flats %.2f pit_paths %.2f terrain %.2f' % (d['ms_per_step'], s['fill_flats_ms'], s['pit_paths_ms'], s['terrain_ms']))"
done; done
PYDEM_FLAT_BATCH=131072 PYDEM_FLAT_BATCH_REGIONS=16382 PYDEM_COND_DEBUG=2 timeout 300 python bench.py --config 5 --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>&1 >/dev/null | grep fill_flats | head -12
