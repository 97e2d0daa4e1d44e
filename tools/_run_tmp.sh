timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/r6/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/r6/tests.log; tail -3 gpurun_out/r6/tests.log
echo "--- bench --gpus 2 without two GPUs:"; timeout 120 python bench.py --gpus 2 --size 2048 --steps 1 --warmup 1 2>&1 | tail -3 | cut -c1-400
echo "--- bench --gpus 2, shared GPU over sockets:"; PYDEM_BENCH_SHARED_GPU=1 timeout 600 python bench.py --gpus 2 --size 4096 --steps 2 --warmup 1 --cpu-sample 0 2> gpurun_out/r6/bench2.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_gpus', 'ms_per_step', 'rccl_ranks', 'per_rank')}); print(d['config']['edge_exchange'], d['sweep'])"
tail -3 gpurun_out/r6/bench2.err
echo "--- bench --gpus 8, shared GPU over sockets:"; PYDEM_BENCH_SHARED_GPU=1 timeout 900 python bench.py --gpus 8 --size 2048 --steps 2 --warmup 1 --cpu-sample 0 2> gpurun_out/r6/bench8.err | python -c "
import sys, json
d = json.loads(sys.stdin.read()); print({k: d[k] for k in ('n_gpus', 'ms_per_step', 'rccl_ranks')}); print([ (p['rank'], p['edge_waves'], p['edge_queued_batches'], p['edge_host_looks']) for p in d['per_rank']])"
tail -3 gpurun_out/r6/bench8.err
timeout 600 python bench.py --config 5 > gpurun_out/r6/bench_config5.json 2>/dev/null; python -c "import json;d=json.load(open('gpurun_out/r6/bench_config5.json'));print(d['ms_per_step'], d['stages_ms']['fill_flats_ms'], d['stages_ms']['pit_paths_ms'])"
PM_WORKERS=8 PM_EDGE_MODE=pool PM_IN_FLIGHT=auto timeout 900 python tools/pm_multitile_timing.py 16384 8 2>&1 | grep "^n=" | tail -1
