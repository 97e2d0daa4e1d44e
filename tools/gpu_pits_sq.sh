#!/bin/bash
# issue statistics of the pit tiers (pair tier on / off)
mkdir -p gpurun_out/psq
cd /tmp && export TMPDIR=/tmp
for pair in 1 0; do
  rm -rf /tmp/psq_$pair
  PYDEM_PITS_PAIR=$pair timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --kernel-include-regex 'k_pits' -d /tmp/psq_$pair -o t --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 0 > /tmp/psq_$pair.log 2>&1
  python $GRAFT_REPO_ROOT/tools/pmc_aggregate.py /tmp/psq_$pair > $GRAFT_REPO_ROOT/gpurun_out/psq/pmc_pair$pair.csv
  echo "== pair=$pair"; cat $GRAFT_REPO_ROOT/gpurun_out/psq/pmc_pair$pair.csv | sed 's/(anonymous namespace):://g' | cut -c1-30,60-200 | grep -v "^kernel"
done
