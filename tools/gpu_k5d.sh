#!/bin/bash
# compact-pass iteration: old vs new schedule, the sweep tests, the bench, the per-pass trace
mkdir -p gpurun_out/k5d
timeout 900 python tools/cmp_sweep_modes.py 1024 4096 > gpurun_out/k5d/cmp.log 2>&1; echo "cmp rc $?" >> gpurun_out/k5d/cmp.log
tail -12 gpurun_out/k5d/cmp.log
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep_modes.py tests/test_gpu_soak.py tests/test_gpu_pits.py -x -q > gpurun_out/k5d/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/k5d/tests.log
tail -4 gpurun_out/k5d/tests.log
PYDEM_SWEEP_DEBUG=1 timeout 600 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 5 > gpurun_out/k5d/bench.json 2> gpurun_out/k5d/bench.err
python - <<'PY'
import json
d=json.load(open('gpurun_out/k5d/bench.json'))
print(d['ms_per_step'], d['stages_ms'], d['sweep'])
PY
grep -c "compact tile pass" gpurun_out/k5d/bench.err; grep "compact passes" gpurun_out/k5d/bench.err | head -2
cd /tmp && export TMPDIR=/tmp && timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -- python $GRAFT_REPO_ROOT/bench.py --steps 1 --warmup 1 --cpu-sample 0 --roof-iters 0 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT && python tools/sweep_passes.py /tmp/tr > gpurun_out/k5d/passes.txt 2>&1; head -30 gpurun_out/k5d/passes.txt; tail -3 gpurun_out/k5d/passes.txt
