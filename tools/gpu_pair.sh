#!/bin/bash
mkdir -p gpurun_out/pair
for occ in 6 5; do
  touch pydem_amd/csrc/pits.hip
  PYDEM_HIPCC_FLAGS="-DPYDEM_PR_OCC=$occ" python -m pydem_amd.build > gpurun_out/pair/build_$occ.log 2>&1 || { tail -5 gpurun_out/pair/build_$occ.log; continue; }
  if [ $occ = 6 ]; then
    timeout 900 python -m pytest tests/test_gpu_pits.py tests/test_gpu_parity.py -x -q > gpurun_out/pair/tests.log 2>&1; grep -n "passed\|failed\|Error\|error" gpurun_out/pair/tests.log | tail -5
  fi
  PYDEM_PITS_DEBUG=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 0 > gpurun_out/pair/bench_$occ.json 2> gpurun_out/pair/bench_$occ.err
  grep "pits:" gpurun_out/pair/bench_$occ.err | tail -2
  python - <<PY
import json
d=json.load(open('gpurun_out/pair/bench_$occ.json'))
print('OCC $occ: ms_per_step %.2f pits_ms %.2f' % (d['ms_per_step'], d['stages_ms']['pits_ms']))
PY
done
timeout 900 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config3" > gpurun_out/pair/large.log 2>&1; grep -n "passed\|failed" gpurun_out/pair/large.log | tail -2
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --roof-iters 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1) && cp $f gpurun_out/pair/kernel_stats.csv && grep "k_pits" gpurun_out/pair/kernel_stats.csv | cut -d, -f1-4 | cut -c1-140
