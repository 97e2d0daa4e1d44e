#!/bin/bash
# the GPU suite on the alternative code paths (each switch is read by the library): cell-indexed incremental rounds, board
# cross-check, exact stencil path, queue schedule of the sweep; since round 6 also: host-built operators, host-driven waves,
# cell-by-cell cascade, no chains in the device build, the stencil's store split, one tile in flight
mkdir -p gpurun_out/alt
run() { name=$1; shift; env "$@" timeout 1500 python -m pytest tests -x -q -m gpu -p no:cacheprovider > gpurun_out/alt/$name.log 2>&1; echo "$name: $(grep -E 'passed|failed' gpurun_out/alt/$name.log | tail -1)"; }
run einc_cell PYDEM_EINC_COMPACT_MAX=0
run board_check PYDEM_BOARD_CHECK=1
run stencil_exact PYDEM_STENCIL_EXACT=1
run sweep_queue PYDEM_SWEEP_MODE=queue
run cond_host PYDEM_COND_BUILD=host
run edge_queue0 PYDEM_EDGE_QUEUE=0
run edge_cond0 PYDEM_EDGE_COND=0
run cb_chain0 PYDEM_CB_CHAIN=0
run stencil_split PYDEM_STENCIL_SPLIT=1
run pits_row_always PYDEM_PITS_ROW=2
run pits_row_never PYDEM_PITS_ROW=0
