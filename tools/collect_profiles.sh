#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench workload, then two separate PMC passes (FETCH_SIZE, WRITE_SIZE).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks -o t --output-format csv -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --roof-iters 5 > gpurun_out/prof/ks.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pf -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 2 > gpurun_out/prof/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pw -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --roof-iters 2 > gpurun_out/prof/pw.log 2>&1
python tools/pmc_aggregate.py gpurun_out/prof/pf gpurun_out/prof/pw > gpurun_out/prof/pmc_fetch_write.csv
cp gpurun_out/prof/ks/t_kernel_stats.csv gpurun_out/prof/kernel_stats.csv
rm -f gpurun_out/prof/*/t_kernel_trace.csv gpurun_out/prof/*/t_counter_collection.csv
head -12 gpurun_out/prof/kernel_stats.csv; head -8 gpurun_out/prof/pmc_fetch_write.csv
