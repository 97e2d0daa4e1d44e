#!/bin/bash
# round-6 driver, one parameterised script: STEPS="bench inflight fixup tests ..." (space-separated), output under gpurun_out/r6/
O=gpurun_out/r6; mkdir -p $O
for s in ${STEPS:-bench}; do
case $s in
bench)    timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; cut -c1-300 $O/bench.json; python -c "import json;d=json.load(open('$O/bench.json'));print(d['ms_per_step'], d['stages_ms'])" ;;
inflight) # two resident 16384^2 tiles: per-tile phases one after the other vs two at a time (what overlapping the stages of two tiles buys)
          for f in 1 2; do PM_IN_FLIGHT=$f timeout 900 python tools/pm_multitile_timing.py ${N:-16384} 2 > $O/inflight_$f.log 2>&1; grep "^n=" $O/inflight_$f.log; done ;;
fixup)    PYDEM_EDGE_DEBUG=1 PYDEM_EDGE_PROFILE=1 PM_WORKERS=8 PM_EDGE_MODE=pool timeout 900 python tools/pm_multitile_timing.py ${N:-16384} 8 > $O/fixup${TAG}.log 2>&1
          grep -v "per wave\|condensed edge round:" $O/fixup${TAG}.log | tail -${TAILN:-30} | cut -c1-420 ;;
tests)    timeout 3000 python -m pytest tests -x -q -m gpu ${PYTEST_ARGS} > $O/tests.log 2>&1; echo "pytest rc $?" >> $O/tests.log; tail -5 $O/tests.log ;;
pmtests)  timeout 1500 python -m pytest tests/test_gpu_process_manager.py tests/test_gpu_edge_update.py -x -q -m gpu ${PYTEST_ARGS} > $O/pmtests.log 2>&1; echo "pytest rc $?" >> $O/pmtests.log; tail -8 $O/pmtests.log ;;
smoke)    timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -2 $O/smoke.log ;;
*)        echo "unknown step $s" ;;
esac
done
