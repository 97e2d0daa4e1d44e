#!/bin/bash
# kernel trace of the queued waves (8 x 16384^2 on one GPU) + config 5 with / without the kept simulations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/queue
rm -rf gpurun_out/queue/kt
PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/queue/kt -o t --output-format csv -- python tools/pm_multitile_timing.py 16384 8 > gpurun_out/queue/kt.log 2>&1
python tools/kstats.py gpurun_out/queue/kt/t_kernel_stats.csv 2>/dev/null | head -40 > gpurun_out/queue/kt_top.txt
python - <<'P' > gpurun_out/queue/kt_edge.txt
import csv
for r in csv.DictReader(open('gpurun_out/queue/kt/t_kernel_stats.csv')):
    n = r['Name']
    if any(k in n for k in ('k_cond', 'k_cb_', 'k_board', 'k_sched', 'k_cinc', 'k_nd_', 'k_einc', 'k_pit_offsets', 'Scan', 'scan', 'sort', 'Sort')):
        short = n.split('(anonymous namespace)::')[-1] if n.startswith('(anonymous') else n
        short = short.replace('(anonymous namespace)::', '')
        print('%-70s %6s calls  total %9.1f us  mean %8.2f us  max %9.1f us' % (short[:70], r['Calls'], int(r['TotalDurationNs']) / 1e3, float(r['AverageNs']) / 1e3, int(r['MaxNs']) / 1e3))
P
cat gpurun_out/queue/kt_edge.txt; grep "edge fix-up wave" gpurun_out/queue/kt.log
rm -f gpurun_out/queue/kt/t_kernel_trace.csv
for keep in $C5_KEEPS; do
PYDEM_PATHS_KEEP=$keep timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null > gpurun_out/queue/c5_keep$keep.json
python -c "
import json; d=json.load(open('gpurun_out/queue/c5_keep$keep.json')); print('keep=$keep', d['ms_per_step'], d['stages_ms']['pit_paths_ms'], d['stages_ms']['pit_path_rounds'], d['stages_ms']['fill_flats_ms'])"
done
