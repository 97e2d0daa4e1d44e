#!/bin/bash
# same-box A/B of two builds of the library: pydem_amd/lib/libpydem_hip.so.A / .B (built beforehand), alternating.
# BENCH_ARGS: extra bench.py arguments (e.g. "--config 5"); KEEP=A|B: which build stays in place afterwards
L=pydem_amd/lib/libpydem_hip.so
for rep in 1 2 3; do
  for v in ${VARIANTS:-A B}; do
    cp $L.$v $L
    timeout 300 python bench.py --steps 3 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 0 ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; print('$v rep $rep: ms_per_step %.2f ' % d['ms_per_step'], ' '.join('%s %.2f' % (k.replace('_ms',''), v) for k, v in s.items() if isinstance(v, (int, float)) and k not in ('slopes_directions_ms','tile_ms','edge_fixup_ms','h2d_ms')))"
  done
done
cp $L.${KEEP:-A} $L
