#!/bin/bash
# config 5: speculation window / large-window batch size of the pit paths (3 steps each: allocation effects show in the mean)
mkdir -p gpurun_out/c5
for cfg in "131072 2048" "131072 1024" "131072 512" "65536 512"; do
  set -- $cfg
  echo "== window $1 big $2"
  PYDEM_PATHS_WINDOW=$1 PYDEM_PATHS_BIG=$2 PYDEM_PATHS_DEBUG=1 timeout 600 python bench.py --config 5 --steps 3 --warmup 1 2>&1 | grep -E "pit drain|ms_per_step" | sed 's/.*"ms_per_step": \([0-9.]*\).*pit_paths_ms": \([0-9.]*\).*/ms_per_step \1 pit_paths_ms(last) \2/' | tail -2
done
