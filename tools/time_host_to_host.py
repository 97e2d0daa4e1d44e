#!/usr/bin/env python
"""Host -> host wall-clock of the drop-in call DEMProcessor(elev=array).calc_twi() next to the device time, per phase:
   time_host_to_host.py [16384 | c5]      (PCIe-inclusive figures: never the bench's `value`)"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
from pydem_amd import DEMProcessor, synth        # noqa: E402

what = sys.argv[1] if len(sys.argv) > 1 else '16384'
if what == 'c5':
    z = synth.srtm_int16(8192, 8192, seed=3)
    kw = {}
else:
    n = int(what)
    z = synth.fractal(n, n, seed=1)
    kw = dict(fill_flats=False, drain_pits_path=False)
for rep in range(3):
    t0 = time.perf_counter()
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, **kw)
    t1 = time.perf_counter()
    dp._ensure_tile(); dp._push('elev'); dp._tile.synchronize()
    t2 = time.perf_counter()
    twi = dp.calc_twi()
    t3 = time.perf_counter()
    tm = dp.timings
    dev = sum(tm[k] for k in ('slopes_directions_ms', 'flats_ms', 'graph_ms', 'pits_ms', 'sweep_ms', 'twi_ms'))
    t4 = time.perf_counter()
    mag = dp.mag; t5 = time.perf_counter(); uca = dp.uca; t6 = time.perf_counter()
    print('%s rep %d: construct %.1f ms, upload %.1f ms, calc_twi() %.1f ms (device stages of the terrain part %.1f ms), then mag %.1f ms, uca %.1f ms; '
          'host -> host (array in, twi out) %.1f ms = %.0f Mcells/s'
          % (what, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3, dev, (t5 - t4) * 1e3, (t6 - t5) * 1e3, (t3 - t0) * 1e3, z.size / (t3 - t0) / 1e6), flush=True)
    del dp, twi, mag, uca
