#!/bin/bash
# wave-tier occupancy / list-capacity variants of the pit search (rebuilds pits.o on the box)
mkdir -p gpurun_out/pocc
for v in "6 256" "8 192" "7 192" "8 256" "8 128"; do
  set -- $v
  touch pydem_amd/csrc/pits.hip
  PYDEM_HIPCC_FLAGS="-DPYDEM_WV_OCC=$1 -DPYDEM_WV_CAP=$2" python -m pydem_amd.build > gpurun_out/pocc/build_$1_$2.log 2>&1 || { echo "build failed $v"; tail -5 gpurun_out/pocc/build_$1_$2.log; continue; }
  PYDEM_PITS_DEBUG=1 timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --roof-iters 0 > gpurun_out/pocc/b_$1_$2.json 2> gpurun_out/pocc/b_$1_$2.err
  python - <<PY
import json
d=json.load(open('gpurun_out/pocc/b_$1_$2.json'))
print('OCC $1 CAP $2: ms_per_step %.2f pits_ms %.2f' % (d['ms_per_step'], d['stages_ms']['pits_ms']))
PY
  grep -i "tier\|overflow\|big" gpurun_out/pocc/b_$1_$2.err | tail -3
done
