#!/bin/bash
mkdir -p gpurun_out/r4g
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_pm && PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_pm -- python $GRAFT_REPO_ROOT/tools/pm_multitile_timing.py 16384 8 > $GRAFT_REPO_ROOT/gpurun_out/r4g/pm.log 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import csv, glob, collections, re
f = sorted(glob.glob('/tmp/tr_pm/**/*kernel_trace.csv', recursive=True))[-1]
rows = [r for r in csv.DictReader(open(f))]
agg = collections.defaultdict(list)
for r in rows:
    nm = r['Kernel_Name']
    if any(k in nm for k in ('k_cond_', 'k_board_', 'k_cinc_', 'k_nd_')):
        agg[re.search(r'k_[a-z_0-9]+', nm).group(0)].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3)
for k, v in sorted(agg.items()):
    v.sort()
    print('%-42s n=%5d  mean %8.1f us  median %8.1f  p90 %8.1f  max %9.1f  sum %9.1f ms' % (k, len(v), sum(v) / len(v), v[len(v) // 2], v[int(len(v) * 0.9)], v[-1], sum(v) / 1e3))
PY
tail -1 gpurun_out/r4g/pm.log | cut -c1-200
