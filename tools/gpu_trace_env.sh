#!/bin/bash
# per-pass durations of the sweep for a given environment:  gpu_trace_env.sh NAME VAR=.. VAR=..
name=$1; shift
mkdir -p gpurun_out/tr
cd /tmp && export TMPDIR=/tmp && rm -rf /tmp/tr_$name && env "$@" timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$name -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/sweep_passes.py /tmp/tr_$name > gpurun_out/tr/$name.txt 2>&1; head -${HEADN:-14} gpurun_out/tr/$name.txt | cut -c1-120; tail -1 gpurun_out/tr/$name.txt
