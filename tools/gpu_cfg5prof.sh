#!/bin/bash
# kernel statistics of the config-5 bench line (conditioning on the device)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/c5
PYDEM_PATHS_DEBUG=1 PYDEM_COND_DEBUG=1 timeout 900 rocprofv3 --kernel-trace --stats -d gpurun_out/c5/ks -o t --output-format csv -- python bench.py --config 5 --steps 1 --warmup 0 > gpurun_out/c5/bench.json 2> gpurun_out/c5/bench.err
cp gpurun_out/c5/ks/t_kernel_stats.csv gpurun_out/c5/kernel_stats.csv
rm -rf gpurun_out/c5/ks
grep -i "fill_flats\|pit drain" gpurun_out/c5/bench.err | head
cat gpurun_out/c5/bench.json | cut -c1-600
