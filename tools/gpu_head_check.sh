#!/bin/bash
# round-end sequence + the config-5 line (HEAD check after a container restart)
bash tools/gpu_full.sh
timeout 600 python bench.py --config 5 > gpurun_out/full/bench_config5.json 2> gpurun_out/full/bench_config5.err; cut -c1-300 gpurun_out/full/bench_config5.json
