#!/usr/bin/env python
"""BASELINE.json config 5: int16 SRTM-style tile with fill_flats=True + drain_pits (all reference defaults).
Reports the host-side conditioning time and the device phases separately.  Usage: run_config5.py [size] [seed]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydem_amd import DEMProcessor, synth        # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 3
t0 = time.perf_counter()
elev = synth.srtm_int16(size, size, seed=seed)
t1 = time.perf_counter()
print("generated %dx%d int16 tile in %.1f s (plateau fraction %.1f %%)" % (size, size, t1 - t0,
      100.0 * np.mean(elev[1:, :] == elev[:-1, :])))
warnings.simplefilter('ignore')
dp = DEMProcessor(elev=elev, dX=30.0, dY=30.0)        # reference defaults: fill_flats, drain_pits_path, drain_pits all on
t2 = time.perf_counter()
# calc_slopes_directions runs the conditioning first (reference :593-599); the stages are timed one by one here
dp.calc_fill_flats()
dp._tile.synchronize() if dp._tile is not None else None
ta = time.perf_counter()
dp.calc_pit_drain_paths()
tb = time.perf_counter()
dp.fill_flats = False
dp.drain_pits_path = False
dp.calc_slopes_directions()
t3 = time.perf_counter()
print("fill_flats (artefacts + flats%s) %.3f s, pit drain paths %.3f s, upload + stencil + download %.3f s"
      % (", device" if 'elev' in dp._on_device or True else "", ta - t2, tb - ta, t3 - tb))
dp.calc_uca()
t4 = time.perf_counter()
twi = dp.calc_twi()
t5 = time.perf_counter()
tm = dp.timings
print("conditioning + stencil: %.2f s" % (t3 - t2))
print("end to end (conditioning, stencil, uca, twi; input on the host): %.2f s = %.1f Mcells/s" % (t5 - t2, size * size / (t5 - t2) / 1e6))
print("calc_uca %.3f s, calc_twi %.3f s (incl. transfers); device stages ms: stencil %.2f flats %.2f graph %.2f pits %.2f sweep %.2f"
      % (t4 - t3, t5 - t4, tm['stencil_kernel_ms'], tm['flats_ms'], tm['graph_ms'], tm['pits_ms'], tm['sweep_ms']))
print("flats left %d, pit edges %d, undrained pits %d, unresolved cells %d, sweep passes %d"
      % (tm['n_flats'], tm['n_pit_edges'], tm['n_pits_undrained'], tm['n_unresolved'], tm.get('sweep_tile_passes', -1)))
uca = dp.uca
ok = np.isfinite(uca) | np.asarray(dp.flats, bool)
print("uca finite outside flats: %s; min %.1f (cell area %.1f); twi finite: %.4f of cells" % (bool(ok.all()), np.nanmin(uca), 900.0,
      np.isfinite(twi).mean()))
