#!/bin/bash
# the GPU suite on the alternative code paths (each switch is read by the library): cell-indexed incremental rounds, board
# cross-check, exact stencil path, queue schedule of the sweep
mkdir -p gpurun_out/alt
run() { name=$1; shift; env "$@" timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/alt/$name.log 2>&1; echo "$name: $(grep -E 'passed|failed' gpurun_out/alt/$name.log | tail -1)"; }
run einc_cell PYDEM_EINC_COMPACT_MAX=0
run board_check PYDEM_BOARD_CHECK=1
run stencil_exact PYDEM_STENCIL_EXACT=1
run sweep_queue PYDEM_SWEEP_MODE=queue
