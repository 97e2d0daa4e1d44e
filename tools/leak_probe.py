#!/usr/bin/env python
"""Free device memory over repeated runs of the paths that allocate: the default DEMProcessor (device conditioning + terrain)
on an SRTM-like int16 tile, created and dropped each time; and a directory run (pool mode, device edge board, RCCL
transport on one rank) created and dropped each time.  A steady state is expected after the first iterations.
    python tools/leak_probe.py [iterations]"""
import gc
import os
import sys
import tempfile
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from pydem_amd import DEMProcessor, _ffi, process_manager, synth   # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 8
warnings.simplefilter('ignore')


def series(name, fn):
    free = []
    for _ in range(N):
        fn()
        gc.collect()
        free.append(_ffi.device_memory(0)[0])
    d = [free[i] - free[i + 1] for i in range(len(free) - 1)]
    print('%-28s free after run 1: %d MiB; shrink per further run (bytes): %s' % (name, free[0] >> 20, d), flush=True)
    return d


def conditioned_tile():
    z = synth.srtm_int16(1536, 1280, seed=3)
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0)
    dp.calc_twi()


def plain_tile():
    z = synth.fractal(1536, 1280, seed=4, top_shift=7, n_octaves=7)
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False)
    dp.calc_twi()


def directory(n_workers):
    def run():
        z = synth.fractal(600, 700, seed=6, top_shift=6, n_octaves=6)
        with tempfile.TemporaryDirectory() as d:
            for t, (elev, bounds) in enumerate(synth.split_mosaic(z, 2, 3, 2)):
                np.savez(os.path.join(d, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
            pm = process_manager.ProcessManager(in_path=d, elev_conditioned=True, n_workers=n_workers, tiles_in_flight=6)
            pm.process_twi()
            pm.save_non_overlap_data()
            if hasattr(pm, 'close'):
                pm.close()
    return run


bad = 0
for name, fn in (('plain tile', plain_tile), ('conditioned int16 tile', conditioned_tile),
                 ('directory, serial order', directory(1)), ('directory, pool mode', directory(8))):
    d = series(name, fn)
    if sum(x > 0 for x in d[2:]) >= 2:          # a leak shrinks the free memory run after run; one late one-off allocation does not count
        bad += 1
print('leak probe:', 'STEADY' if bad == 0 else '%d scenario(s) keep shrinking' % bad)
sys.exit(1 if bad else 0)
