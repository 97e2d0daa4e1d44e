#!/bin/bash
# kernel trace of the pit search (bench tile, 16384^2): one line per k_pits_* kernel; PITS_ENVS = list of "name:VAR=val,VAR=val" variants
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/pits
for spec in ${PITS_ENVS:-default:PYDEM_PITS_ROW=1 norow:PYDEM_PITS_ROW=0}; do
  name=${spec%%:*}; envs=$(echo "${spec#*:}" | tr ',' ' ')
  rm -rf gpurun_out/pits/kt_$name
  env $envs timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/pits/kt_$name -o t --output-format csv -- python bench.py --steps ${PITS_STEPS:-4} --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 > gpurun_out/pits/kt_$name.log 2>&1
  python - "$name" <<'P'
import csv, sys
name = sys.argv[1]
tot = 0.0
for r in csv.DictReader(open('gpurun_out/pits/kt_%s/t_kernel_stats.csv' % name)):
    n = r['Name'].replace('(anonymous namespace)::', '')
    if 'k_pits' in n or 'k_pit' in n or 'k_compact_mask' in n:
        print('%-10s %-40s %4s calls  mean %9.2f us  max %9.1f us' % (name, n[:40], r['Calls'], float(r['AverageNs']) / 1e3, int(r['MaxNs']) / 1e3))
P
  grep -o '"ms_per_step": [0-9.]*' gpurun_out/pits/kt_$name.log | head -1
  rm -f gpurun_out/pits/kt_$name/t_kernel_trace.csv
done
