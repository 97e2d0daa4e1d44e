#!/bin/bash
# SQ counters of the stencil kernel (one pass, restricted to k_stencil*)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sq
rocprofv3 -L > gpurun_out/sq/avail.txt 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_WAVES --kernel-include-regex 'k_stencil_march' -d gpurun_out/sq/a -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 2 > gpurun_out/sq/a.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_MISC SQ_WAIT_INST_LDS --kernel-include-regex 'k_stencil_march' -d gpurun_out/sq/b -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 2 > gpurun_out/sq/b.log 2>&1
python tools/pmc_aggregate.py gpurun_out/sq/a gpurun_out/sq/b > gpurun_out/sq/pmc_sq.csv
rm -rf gpurun_out/sq/a gpurun_out/sq/b
cut -d, -f2- gpurun_out/sq/pmc_sq.csv | grep -v "^counter" | sed 's/.*)",//' 
tail -3 gpurun_out/sq/b.log
