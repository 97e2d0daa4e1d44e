#!/bin/bash
# per-pass durations of the tile sweep (kernel trace of one bench step) + the list sizes the host sees
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/sw
PYDEM_SWEEP_DEBUG=1 timeout 600 rocprofv3 --kernel-trace -d gpurun_out/sw/tr -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 1 > gpurun_out/sw/bench.json 2> gpurun_out/sw/bench.err
python tools/sweep_passes.py gpurun_out/sw/tr 1 > gpurun_out/sw/passes.txt
grep "listed\|tile passes" gpurun_out/sw/bench.err | head -30 > gpurun_out/sw/lists.txt
rm -rf gpurun_out/sw/tr
tail -4 gpurun_out/sw/passes.txt
