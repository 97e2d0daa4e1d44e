#!/bin/bash
# the driver's round-end sequence: GPU tests, smoke, default bench
mkdir -p gpurun_out/full
timeout 3000 python -m pytest tests -x -q -m gpu > gpurun_out/full/tests.log 2>&1; echo "pytest rc $?" >> gpurun_out/full/tests.log
tail -5 gpurun_out/full/tests.log
timeout 600 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/full/smoke.log 2>&1; tail -2 gpurun_out/full/smoke.log
timeout 900 python bench.py > gpurun_out/full/bench.json 2> gpurun_out/full/bench.err; cat gpurun_out/full/bench.json | cut -c1-400
