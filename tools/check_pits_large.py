#!/usr/bin/env python
"""Compare the device pit -> drain assignment with the CPU oracle on one large fractal tile
(the bench generator), where the rare large-basin pits appear.  Usage: check_pits_large.py [size] [seed]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O            # noqa: E402  (checker only)
from pydem_amd import DEMProcessor        # noqa: E402

size = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
if len(sys.argv) > 3 and sys.argv[3] == 'srtm':
    # config-5 style surface: int16 plateaus, conditioned on the host (fill_flats + pit drain paths), then the pit solver
    # meets borders of hundreds of cells and many pits that never drain
    from pydem_amd import synth, conditioning
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        z = conditioning.fill_flats(synth.srtm_int16(size, size, seed=seed))
        z, _, _ = conditioning.pit_drain_paths(z, 30.0 * np.ones(size - 1), 30.0 * np.ones(size - 1))
else:
    z = O.synth_fractal(size, size, seed=seed)
t0 = time.time()
o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
o.calc_uca()
print("oracle %.1f s: %d pit edges, %d undrained" % (time.time() - t0, len(o.pit_i), o.n_warn))
dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    dp.calc_slopes_directions()
    dp.calc_uca()
src, dst, w = dp._tile.pit_edges()
print("device: %d pit edges, %d undrained" % (len(src), dp.timings['n_pits_undrained']))
ref = sorted(zip(o.pit_i.tolist(), o.pit_j.tolist()))
got = sorted(zip(src.tolist(), dst.tolist()))
if ref != got:
    rs, gs = set(ref), set(got)
    print("MISMATCH: %d only in oracle, %d only on device" % (len(rs - gs), len(gs - rs)))
    print(" oracle-only pits:", sorted({a for a, _ in rs - gs})[:10])
    print(" device-only pits:", sorted({a for a, _ in gs - rs})[:10])
    sys.exit(1)
wr = np.array([x[2] for x in sorted(zip(o.pit_i.tolist(), o.pit_j.tolist(), o.pit_prop.tolist()))])
wg = np.array([x[2] for x in sorted(zip(src.tolist(), dst.tolist(), w.tolist()))])
print("assignments identical; max rel weight diff %.3g" % np.max(np.abs(wr - wg) / np.abs(wr)))
assert dp.timings['n_pits_undrained'] == o.n_warn
np.testing.assert_allclose(dp.uca, o.uca, rtol=1e-9)
print("uca matches")
