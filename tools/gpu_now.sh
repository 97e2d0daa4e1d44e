#!/bin/bash
# short loop of the running session: bench line, kernel statistics of one step, the pit search's debug line (output under gpurun_out/now/)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/now; rm -rf $O; mkdir -p $O
timeout 600 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-400 $O/bench.json
python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['stages_ms'])"
timeout 600 rocprofv3 --kernel-trace --stats -d $O/ks -o t --output-format csv -- python bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 2 > $O/ks.log 2>&1
cp $O/ks/t_kernel_stats.csv $O/kernel_stats.csv; python tools/sweep_passes.py $O/ks/t_kernel_trace.csv > $O/sweep_passes.txt 2>&1
rm -f $O/ks/t_kernel_trace.csv
head -30 $O/kernel_stats.csv | cut -c1-200
PYDEM_PITS_DEBUG=1 timeout 600 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 > $O/pitsdbg.json 2> $O/pitsdbg.err; tail -12 $O/pitsdbg.err | cut -c1-400
