#!/bin/bash
# Issue counters of the conditioning kernels (config 5: pit drain paths, flats): SQ passes only (the TA / TCP / TCC sets of
# tools/gpu_pmc_hot.sh exceed what one pass can collect on this device)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_c5; rm -rf $O; mkdir -p $O
RX='k_paths_small|k_paths_big|k_flat_batch|k_pits_block|k_pits_wave_big|k_paths_commit|k_paths_tentative'
BENCH="python bench.py --config 5 --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 0"
i=0
while read -r SET; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $SET --kernel-include-regex "$RX" -d $O/p$i -o t --output-format csv -- $BENCH > $O/p$i.log 2>&1 || echo "pass $i ($SET) failed" >> $O/errors.txt
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_WAIT_ANY SQ_LDS_BANK_CONFLICT
SETS
python tools/pmc_sum.py $O/p* > $O/table.csv
rm -rf $O/p*/
wc -l $O/table.csv; cat $O/errors.txt 2>/dev/null
