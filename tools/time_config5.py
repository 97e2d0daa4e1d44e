#!/usr/bin/env python
"""GPU box: wall-clock of the stages of BASELINE config 5 (8192^2 int16, reference defaults) around the C-ABI calls,
next to the library's own stage timers: finds host-side time the stage timers do not see."""
import sys, time, warnings, os
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
warnings.simplefilter('ignore')
import numpy as np
from pydem_amd import DEMProcessor, synth
n = 8192
z = synth.srtm_int16(n, n, seed=3)
dp = DEMProcessor(elev=z, dX=30.0, dY=30.0)
for it in range(4):
    dp.elev = z
    dp._ensure_tile(); dp._push('elev'); dp._tile.synchronize()
    t1 = time.perf_counter()
    dp.fill_flats = True; dp.drain_pits_path = True
    dp.calc_fill_flats(); dp._tile.synchronize()
    t2 = time.perf_counter()
    r = dp._pit_paths_on_device()
    t3 = time.perf_counter()
    dp._tile.synchronize()
    t4 = time.perf_counter()
    dp.fill_flats = False; dp.drain_pits_path = False
    time.sleep(0.05 * (it % 2)); tz = time.perf_counter(); dp._tile.get_line(0, 0, 5); print('   tiny gather + D2H before slopes: %.2f ms' % ((time.perf_counter() - tz) * 1e3))
    ta = time.perf_counter(); dp._ensure_tile(); tb = time.perf_counter(); dp._push('elev'); tc = time.perf_counter()
    dp._tile.slopes_directions(); td = time.perf_counter(); dp._tile.slopes_directions(); te = time.perf_counter()
    print('   ensure_tile %.2f push %.2f slopes call %.2f again %.2f' % ((tb - ta) * 1e3, (tc - tb) * 1e3, (td - tc) * 1e3, (te - td) * 1e3))
    dp.run_slopes_directions(); dp._tile.synchronize(); t5 = time.perf_counter()
    dp.run_uca(); dp._tile.synchronize(); t6 = time.perf_counter()
    dp.run_twi(); dp._tile.synchronize(); t7 = time.perf_counter()
    print('fill_flats %.1f ms, pit paths call %.1f ms, sync after %.1f ms; slopes %.1f uca %.1f twi %.1f' % ((t2 - t1) * 1e3, (t3 - t2) * 1e3, (t4 - t3) * 1e3,
          (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t7 - t6) * 1e3), r, {k: round(v, 2) for k, v in dp._tile.timings().items() if k.endswith('_ms')})
