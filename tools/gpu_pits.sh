#!/bin/bash
# pit-search iteration: the pit tests, then the bench line and the kernel times of the pit tiers
mkdir -p gpurun_out/pits
timeout 900 python -m pytest tests/test_gpu_pits.py tests/test_gpu_parity.py -x -q > gpurun_out/pits/tests.log 2>&1; grep -n "passed\|failed" gpurun_out/pits/tests.log | tail -2
timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 0 > gpurun_out/pits/bench.json 2> gpurun_out/pits/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/pits/bench.json'))
print('ms_per_step %.2f' % d['ms_per_step'], {k: round(v, 2) for k, v in d['stages_ms'].items()})
PY
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/pk && timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -o t -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --roof-iters 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && f=$(find /tmp/pk -name "*kernel_stats.csv" | head -1) && cp $f gpurun_out/pits/kernel_stats.csv && python - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/pits/kernel_stats.csv')))
for r in rows[:16]:
    print('%-60s calls %4s total %9.3f ms avg %9.3f ms' % (r['Name'].replace('(anonymous namespace)::','')[:60], r['Calls'], float(r['TotalDurationNs'])/1e6, float(r['AverageNs'])/1e6))
PY
