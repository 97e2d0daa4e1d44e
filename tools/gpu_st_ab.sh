#!/bin/bash
# same-box A/B of library variants on the stencil only (config 2 = 4096^2 stencil kernel, and the 16384^2 back-to-back figure)
L=pydem_amd/lib/libpydem_hip.so
for rep in 1 2; do
  for v in ${VARIANTS:-A B}; do
    cp $L.$v $L
    timeout 300 python bench.py --steps 2 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 10 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline_stencil']; print('$v rep $rep: in the pipeline %.3f ms, back to back %.3f ms' % (r['avg_kernel_ms'], r['back_to_back_ms']))"
  done
done
cp $L.${KEEP:-A} $L
