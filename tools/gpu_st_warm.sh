#!/bin/bash
# what does the stencil pay for being the first kernel of a step?  (PYDEM_STENCIL_WARM, csrc/stencil.hip)
for w in 0 1 2 3 0 1; do
PYDEM_STENCIL_WARM=$w timeout 300 python bench.py --steps 4 --warmup 1 --cpu-sample 0 --host-to-host 0 --roof-iters 8 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); s=d['stages_ms']; r=d['roofline_stencil']; print('warm=$w: stencil in the pipeline %.3f ms, back to back %.3f ms' % (s['stencil_kernel_ms'], r.get('back_to_back_ms', -1)))"
done
