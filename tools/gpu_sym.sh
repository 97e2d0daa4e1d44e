#!/bin/bash
# two-level solve of the sweep (K5f): parity tests with it on, statistics, A/B of the bench line over switch points
# (PYDEM_SWEEP_SYM = listed tiles at or below which the symbolic pass takes over; 0 = tile passes only)
mkdir -p gpurun_out/sym
PYDEM_SWEEP_SYM=${SYM_TEST:-100000000} timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_soak.py tests/test_gpu_pits.py -x -q > gpurun_out/sym/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/sym/tests.log
tail -3 gpurun_out/sym/tests.log
for rep in 1 2; do
for v in ${SYM_LIST:-0 4096 32768 100000000}; do
PYDEM_SWEEP_SYM=$v timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('sym=$v rep $rep: ms_per_step %.2f ' % d['ms_per_step'], ' '.join('%s %.2f' % (k.replace('_ms',''), v) for k, v in s.items() if isinstance(v, (int, float)) and k not in ('slopes_directions_ms','tile_ms','edge_fixup_ms','h2d_ms')))"
done
done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for v in ${SYM_TRACE:-4096 100000000}; do
PYDEM_SWEEP_SYM=$v PYDEM_SWEEP_DEBUG=1 timeout 600 rocprofv3 --kernel-trace -d gpurun_out/sym/tr -o t --output-format csv -- python bench.py --steps 1 --warmup 0 --cpu-sample 0 --host-to-host 0 --roof-iters 0 > gpurun_out/sym/bench_$v.json 2> gpurun_out/sym/bench_$v.err
python tools/sweep_passes.py gpurun_out/sym/tr 1 > gpurun_out/sym/passes_$v.txt
grep "listed\|tile passes\|two-level\|symbolic" gpurun_out/sym/bench_$v.err > gpurun_out/sym/lists_$v.txt
rm -rf gpurun_out/sym/tr
tail -1 gpurun_out/sym/passes_$v.txt; grep "k_sweep_sym\|k_sym_fin" gpurun_out/sym/passes_$v.txt; grep "two-level\|symbolic" gpurun_out/sym/lists_$v.txt
done
PYDEM_SWEEP_SYM=${SYM_TEST:-100000000} timeout 900 python -m pytest tests/test_gpu_large_configs.py -x -q -k "config3" > gpurun_out/sym/large.log 2>&1; tail -3 gpurun_out/sym/large.log
