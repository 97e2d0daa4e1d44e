#!/bin/bash
# issue statistics of the sweep kernels
mkdir -p gpurun_out/ssq
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ssq
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY --kernel-include-regex 'k_sweep' -d /tmp/ssq -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 0 > /tmp/ssq.log 2>&1
python $GRAFT_REPO_ROOT/tools/pmc_aggregate.py /tmp/ssq > $GRAFT_REPO_ROOT/gpurun_out/ssq/pmc_sweep.csv
python - <<'PY'
import csv, os
rows=list(csv.DictReader(open(os.environ['GRAFT_REPO_ROOT']+'/gpurun_out/ssq/pmc_sweep.csv')))
for r in rows:
    k=r['kernel'].replace('(anonymous namespace)::','')[:40]
    print('%-42s %-18s disp %4s mean %14.0f total %14.0f' % (k, r['counter'], r['dispatches'], float(r['mean_per_dispatch_KB']), float(r['mean_per_dispatch_KB'])*int(r['dispatches'])))
PY
