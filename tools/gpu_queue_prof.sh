#!/bin/bash
# kernel trace of the queued waves (8 x 16384^2 on one GPU) + config 5 with / without the kept simulations
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/queue
rm -rf gpurun_out/queue/kt
PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/queue/kt -o t --output-format csv -- python tools/pm_multitile_timing.py 16384 8 > gpurun_out/queue/kt.log 2>&1
python tools/kstats.py gpurun_out/queue/kt/t_kernel_stats.csv 2>/dev/null | head -40 > gpurun_out/queue/kt_top.txt
python - <<'P' > gpurun_out/queue/kt_edge.txt
import csv, collections
rows = list(csv.DictReader(open('gpurun_out/queue/kt/t_kernel_trace.csv')))
agg = collections.defaultdict(lambda: [0, 0.0])
for r in rows:
    n = r['Kernel_Name'].split('(')[0]
    if any(k in n for k in ('k_cond', 'k_board', 'k_sched', 'k_cinc', 'k_nd_', 'k_einc', 'k_pit_offsets')):
        d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) * 1e-3
        agg[n][0] += 1; agg[n][1] += d
for n, (c, us) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    print('%-60s %7d launches %10.1f us total %8.2f us mean' % (n[:60], c, us, us / c))
P
cat gpurun_out/queue/kt_edge.txt
rm -f gpurun_out/queue/kt/t_kernel_trace.csv
for keep in 1 0; do
PYDEM_PATHS_KEEP=$keep timeout 300 python bench.py --config 5 --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null > gpurun_out/queue/c5_keep$keep.json
python -c "
import json; d=json.load(open('gpurun_out/queue/c5_keep$keep.json')); print('keep=$keep', d['ms_per_step'], d['stages_ms']['pit_paths_ms'], d['stages_ms']['pit_path_rounds'], d['stages_ms']['fill_flats_ms'])"
done
