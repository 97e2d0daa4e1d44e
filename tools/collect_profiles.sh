#!/bin/bash
# Collect the rocprofv3 evidence for profiles/ on the GPU box (run through gpurun from the repo root):
#   kernel-trace stats of the default bench workload, two separate PMC passes (FETCH_SIZE, WRITE_SIZE), one SQ pass
#   restricted to the stencil kernel, the bench lines of the other stated configs.  Copy what is to be judged from
#   gpurun_out/prof/ into profiles/ (named per round).
set -x
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/prof && mkdir -p gpurun_out/prof
timeout 600 rocprofv3 --kernel-trace --stats -d gpurun_out/prof/ks -o t --output-format csv -- python bench.py --steps 1 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 5 > gpurun_out/prof/ks.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/prof/pf -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 2 > gpurun_out/prof/pf.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/prof/pw -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 2 > gpurun_out/prof/pw.log 2>&1
python tools/pmc_aggregate.py gpurun_out/prof/pf gpurun_out/prof/pw > gpurun_out/prof/pmc_fetch_write.csv
python tools/csrc_stamp.py > gpurun_out/prof/pmc_fetch_write.meta.json      # identity of the kernels the counters belong to (bench.py: pmc_traffic)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-include-regex 'k_stencil' -d gpurun_out/prof/sq -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 2 > gpurun_out/prof/sq.log 2>&1
python tools/pmc_aggregate.py gpurun_out/prof/sq > gpurun_out/prof/pmc_sq_stencil.csv
cp gpurun_out/prof/pmc_fetch_write.meta.json gpurun_out/prof/pmc_sq_stencil.meta.json      # (bench.py: stencil_valu_insts)
cp gpurun_out/prof/ks/t_kernel_stats.csv gpurun_out/prof/kernel_stats.csv
rm -f gpurun_out/prof/*/t_kernel_trace.csv gpurun_out/prof/*/t_counter_collection.csv
head -14 gpurun_out/prof/kernel_stats.csv; head -8 gpurun_out/prof/pmc_fetch_write.csv; cat gpurun_out/prof/pmc_sq_stencil.csv
# the bench lines quote `traffic` from the committed PMC files while their stamp matches the kernel sources: file the fresh ones first
R=${ROUND_TAG:-r06}
cp gpurun_out/prof/pmc_fetch_write.csv profiles/${R}_pmc_fetch_write_16384.csv; cp gpurun_out/prof/pmc_fetch_write.meta.json profiles/${R}_pmc_fetch_write_16384.meta.json
cp gpurun_out/prof/pmc_sq_stencil.csv profiles/${R}_pmc_sq_stencil_16384.csv; cp gpurun_out/prof/pmc_sq_stencil.meta.json profiles/${R}_pmc_sq_stencil_16384.meta.json
timeout 600 python bench.py > gpurun_out/prof/bench_default.json 2> gpurun_out/prof/bench_default.err
timeout 600 python bench.py --drain-pits 0 --cpu-sample 0 --host-to-host 0 > gpurun_out/prof/bench_nopits.json 2> gpurun_out/prof/bench_nopits.err
timeout 300 python bench.py --config 2 > gpurun_out/prof/bench_config2.json 2> gpurun_out/prof/bench_config2.err
timeout 600 python bench.py --config 5 > gpurun_out/prof/bench_config5.json 2> gpurun_out/prof/bench_config5.err
cat gpurun_out/prof/bench_*.json
# the fix-up of 8 tiles on one GPU: the default (waves chosen on the device, queued 16 at a time) with the host's wall clock by part;
# the host-driven wave loop with every round waited for (per-round times); the round-3 cascade; no waiting; the reference's serial order
PYDEM_EDGE_PROFILE=1 PYDEM_EDGE_DEBUG=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 2>&1 | grep -v "condensed edge round: [0-9]" > gpurun_out/prof/pm_pool_8x16384_queued.log
PYDEM_EDGE_QUEUE=0 PYDEM_EDGE_PROFILE=1 PYDEM_EDGE_SYNC=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/prof/pm_pool_8x16384.log 2>&1
PYDEM_EDGE_QUEUE=0 PYDEM_EDGE_COND=0 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/prof/pm_pool_8x16384_cell_by_cell.log 2>&1
PYDEM_EDGE_QUEUE=0 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/prof/pm_pool_8x16384_async.log 2>&1
PM_WORKERS=1 PM_EDGE_MODE=reference timeout 900 python tools/pm_multitile_timing.py 16384 8 > gpurun_out/prof/pm_serial_8x16384.log 2>&1
grep -v "per wave" gpurun_out/prof/pm_pool_8x16384_queued.log | tail -5 | cut -c1-300; tail -2 gpurun_out/prof/pm_pool_8x16384.log | cut -c1-200; tail -2 gpurun_out/prof/pm_serial_8x16384.log | cut -c1-200
