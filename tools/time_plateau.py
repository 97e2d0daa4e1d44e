#!/usr/bin/env python
"""pits / sweep stage times on a config-5 style tile (int16 plateaus, conditioned on the host):  time_plateau.py [size]"""
import os
import sys
import warnings
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydem_amd import DEMProcessor, synth, conditioning
size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    z = conditioning.fill_flats(synth.srtm_int16(size, size, seed=1))
    z, _, _ = conditioning.pit_drain_paths(z, 30.0 * np.ones(size - 1), 30.0 * np.ones(size - 1))
    for rep in range(2):
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
        dp.calc_slopes_directions(); dp.calc_uca()
        tm = dp.timings
print('size %d: pits %.2f ms, sweep %.2f ms, pit edges %d, undrained %d' % (size, tm['pits_ms'], tm['sweep_ms'], tm['n_pit_edges'], tm['n_pits_undrained']))
