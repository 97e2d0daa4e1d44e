#!/bin/bash
# first-pass experiments: LDS-resident pass 1 (K5a) with a round limit / without chain walking, generic passes afterwards
mkdir -p gpurun_out/first
run() {
  echo "== $*"
  env "$@" PYDEM_SWEEP_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --roof-iters 0 > gpurun_out/first/b.json 2> gpurun_out/first/b.err
  grep "tile passes 1-2" gpurun_out/first/b.err | tail -1
  python - <<'PY'
import json
d=json.load(open('gpurun_out/first/b.json'))
print('   ms_per_step %.2f sweep_ms %.2f passes %s' % (d['ms_per_step'], d['stages_ms']['sweep_ms'], d['sweep'].get('kernel_launches')))
PY
}
run PYDEM_SWEEP_COMPACT=0
run PYDEM_SWEEP_COMPACT=1
for R in 6 10 14 20 1000; do
  run PYDEM_SWEEP_COMPACT=0 PYDEM_SWEEP_FIRST=lds PYDEM_FIRST_CHAIN=0 PYDEM_FIRST_ROUNDS=$R
done
run PYDEM_SWEEP_COMPACT=0 PYDEM_SWEEP_FIRST=lds PYDEM_FIRST_CHAIN=1
