#!/bin/bash
# config 5: kernel stats, sweeps per pass 8 (B) / 16 (D), issue counters of the conditioning kernels
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r4r; mkdir -p $O
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks -o t --output-format csv -- python bench.py --config 5 --steps 2 --warmup 1 --cpu-sample 0 --host-to-host 0 > $O/ks.log 2>&1
cp $O/ks/t_kernel_stats.csv $O/kernel_stats_config5.csv; rm -rf $O/ks
head -25 $O/kernel_stats_config5.csv | cut -c1-150
VARIANTS="B D" KEEP=B BENCH_ARGS="--config 5" bash tools/gpu_ab_lib.sh 2>&1 | tee $O/ab.log
cp pydem_amd/lib/libpydem_hip.so.D pydem_amd/lib/libpydem_hip.so
timeout 600 python -m pytest tests/test_gpu_conditioning.py -x -q 2>&1 | tail -2
timeout 200 python tools/soak_conditioning_device.py 60 2>&1 | tail -1
cp pydem_amd/lib/libpydem_hip.so.B pydem_amd/lib/libpydem_hip.so
bash tools/gpu_pmc_c5.sh
