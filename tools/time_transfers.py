#!/usr/bin/env python
"""Whole-plane transfers between pageable numpy arrays and a tile (pydem_tile_upload / pydem_tile_download):
   time_transfers.py [n]   -- fresh arrays and reused ones, float64 and int16; run with PYDEM_XFER_THREADS=0 for the plain copy"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydem_amd import DEMProcessor, synth, _ffi        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
z0 = synth.fractal(n, n, seed=1)
print('PYDEM_XFER_THREADS=%s' % os.environ.get('PYDEM_XFER_THREADS', '(default)'), flush=True)
for dtype in (np.float64, np.int16):
    for rep in range(3):
        z = np.array(z0, dtype=dtype)                      # a fresh pageable array each time
        t0 = time.perf_counter()
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False)
        dp._ensure_tile(); dp._push('elev'); dp._tile.synchronize()
        t1 = time.perf_counter()
        dp._on_device.discard('elev'); dp._push('elev'); dp._tile.synchronize()
        t2 = time.perf_counter()
        back = dp._tile.download(_ffi.ELEV)
        t3 = time.perf_counter()
        ok = back is not None and np.array_equal(np.asarray(back).reshape(n, n), z.astype(np.float64))
        gb = z.nbytes / 1e9
        print('%s rep %d: first upload (with tile creation) %.1f ms, same array again %.1f ms = %.1f GB/s; download of the float64 plane into a fresh '
              'array %.1f ms = %.1f GB/s; round trip equal: %s' % (np.dtype(dtype).name, rep, (t1 - t0) * 1e3, (t2 - t1) * 1e3, gb / (t2 - t1),
                                                                  (t3 - t2) * 1e3, n * n * 8 / 1e9 / (t3 - t2), ok), flush=True)
        del dp, z, back
