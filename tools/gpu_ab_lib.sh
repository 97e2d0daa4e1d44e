#!/bin/bash
# same-box A/B of two builds of the library: pydem_amd/lib/libpydem_hip.so.A / .B (built beforehand), alternating
L=pydem_amd/lib/libpydem_hip.so
for rep in 1 2 3; do
  for v in A B; do
    cp $L.$v $L
    timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --roof-iters 0 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$v rep $rep: ms_per_step %.2f  sweep %.2f  pits %.2f  stencil %.3f  b2b %s' % (d['ms_per_step'], s.get('sweep_ms',0), s.get('pits_ms',0), s.get('stencil_kernel_ms',0), (d.get('roofline_stencil') or {}).get('back_to_back_ms')))"
  done
done
cp $L.${KEEP:-A} $L
