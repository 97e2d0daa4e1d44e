#!/bin/bash
# Issue / memory-pipeline counters of the latency-bound kernels (tile passes of the sweep, pit search): separate --pmc passes
# of the default bench workload (one step), kernel-filtered.  Run through gpurun; output gpurun_out/pmc_hot/table.csv
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/pmc_hot; rm -rf $O; mkdir -p $O
RX='k_sweep_tiles|k_pits_wave|k_pits_lane|k_stencil_march'
BENCH="python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 0"
i=0
while read -r SET; do
  i=$((i+1))
  timeout 400 rocprofv3 --pmc $SET --kernel-include-regex "$RX" -d $O/p$i -o t --output-format csv -- $BENCH > $O/p$i.log 2>&1 || echo "pass $i ($SET) failed" >> $O/errors.txt
done <<'SETS'
SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS
SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ATOMIC_RETURN SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_SMEM SQ_WAIT_ANY
TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_LOAD_WAVEFRONTS_sum TA_FLAT_STORE_WAVEFRONTS_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum
TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum TCP_TOTAL_ACCESSES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_TCR_TCP_STALL_CYCLES_sum
TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_sum TCC_TAG_STALL_sum TCC_BUSY_sum TCC_EA0_RDREQ_32B_sum
GRBM_GUI_ACTIVE GRBM_COUNT
SETS
python tools/pmc_sum.py $O/p* > $O/table.csv
rm -rf $O/p*/
wc -l $O/table.csv; cat $O/errors.txt 2>/dev/null
