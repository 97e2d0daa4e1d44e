#!/bin/bash
# phases in the two full passes only (B, C = 2 x 2 phases) against one launch per pass (A): six alternating runs
L=pydem_amd/lib/libpydem_hip.so
for rep in 1 2 3 4 5 6; do
  for v in A B C; do
    cp $L.$v $L
    timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$v rep $rep: step %.2f sweep %.2f' % (d['ms_per_step'], d['stages_ms']['sweep_ms']))"
  done
done
cp $L.B $L
