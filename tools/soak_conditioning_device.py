#!/usr/bin/env python
"""Randomised soak of the conditioning ON THE DEVICE (pydem_fill_flats, pydem_pit_paths through DEMProcessor.calc_fill_flats /
calc_pit_drain_paths) against the host implementation (pydem_amd/conditioning.py, itself bit-exact against the reference's
goldens and the numpy restatements): float64 tiles with quantised plateaus, lakes, summit flats, flats on the tile edge, sea
level, every option of the two steps.  Exact equality of the conditioned surface after each step; stops at the first
mismatch and prints the recipe.   soak_conditioning_device.py [seconds] [first_case]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pydem_amd import DEMProcessor, conditioning, synth    # noqa: E402


def make_case(k):
    rng = np.random.default_rng(9000 + k)
    n, m = int(rng.integers(3, 420)), int(rng.integers(3, 420))
    if rng.random() < 0.2:
        n, m = int(rng.integers(3, 30)), int(rng.integers(3, 30))
    if os.environ.get('SOAK_BIG') == '1':
        n, m = int(rng.integers(700, 1800)), int(rng.integers(700, 1800))
    ts = int(rng.integers(2, 8))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -20.0])), zrange=float(rng.choice([600.0, 80.0, 15.0, 4.0])))
    kind = rng.random()
    if kind < 0.45:
        z = np.rint(z)                                              # integer heights: quantisation pits, plateaus, ties
    elif kind < 0.6:
        z = np.rint(z * 2) / 2
    if rng.random() < 0.4:                                          # lakes: flood to an exact level
        lvl = float(np.quantile(z, rng.choice([0.2, 0.5, 0.8])))
        lvl = float(np.rint(lvl)) if kind < 0.45 else lvl
        z = np.where(z < lvl, lvl, z)
    if rng.random() < 0.3:
        z[z < 0] = 0.0                                              # sea
    if os.environ.get('SOAK_NAN') == '1' and rng.random() < 0.85:   # no-data: blocks, a margin, scattered cells, isolated cells in a lake
        z = np.array(z, np.float64)
        kindn = rng.random()
        if kindn < 0.4:
            for _ in range(int(rng.integers(1, 5))):
                i0, j0 = int(rng.integers(0, n)), int(rng.integers(0, m))
                z[i0:i0 + int(rng.integers(1, max(2, n // 3))), j0:j0 + int(rng.integers(1, max(2, m // 3)))] = np.nan
        elif kindn < 0.6:
            z[:, : int(rng.integers(1, max(2, m // 4)))] = np.nan
            if rng.random() < 0.5:
                z[-int(rng.integers(1, max(2, n // 4))):, :] = np.nan
        elif kindn < 0.85:
            z[rng.random(z.shape) < float(rng.choice([0.002, 0.02, 0.15]))] = np.nan
        else:
            z[z <= np.nanquantile(z, 0.3)] = np.nan                # the sea as no-data
    opts = dict(fill_flats_below_sea=bool(rng.random() < 0.3), fill_flats_source_tol=float(rng.choice([1, 0.5, 3])),
                fill_flats_peaks=bool(rng.random() < 0.7), fill_flats_pits=bool(rng.random() < 0.7),
                maximum_pit_area=float(rng.choice([32.0, 0.0, 4.0, 400.0])),
                drain_pits_max_iter=int(rng.choice([300, 40, 5])), drain_pits_max_dist=int(rng.choice([32, 6, 0])),
                drain_pits_max_dist_XY=(None if rng.random() < 0.7 else float(rng.choice([90.0, 400.0]))))
    dX = float(rng.choice([30.0, 1.0, 12.5])); dY = float(rng.choice([30.0, 1.0, 25.0]))
    return dict(case=k, shape=(n, m), options=opts, dX=dX, dY=dY), np.ascontiguousarray(z, np.float64), opts, dX, dY


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); done = 0; host_fallbacks = 0
    warnings.simplefilter('ignore')
    while time.time() - t0 < budget:
        rec, z, o, dX, dY = make_case(k)
        k += 1
        n = z.shape[0]
        want1 = conditioning.fill_flats(z, o['maximum_pit_area'], o['fill_flats_below_sea'], o['fill_flats_source_tol'],
                                        o['fill_flats_peaks'], o['fill_flats_pits'])
        want2, _, used = conditioning.pit_drain_paths(want1.copy(), np.full(n - 1, dX), np.full(n - 1, dY), o['drain_pits_max_iter'],
                                                      o['drain_pits_max_dist'], o['drain_pits_max_dist_XY'], o['fill_flats_below_sea'])
        dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, **o)
        dp.calc_fill_flats()
        if os.environ.get('SOAK_NAN') == '1' and 'elev' not in dp._on_device:
            host_flats = globals().get('host_flats', 0) + 1; globals()['host_flats'] = host_flats
        got1 = np.array(dp.elev)
        if not np.array_equal(got1, want1, equal_nan=True):
            print('MISMATCH after fill_flats', rec, int((got1 != want1).sum()), 'cells'); sys.exit(1)
        dp.calc_pit_drain_paths()
        got2 = np.array(dp.elev)
        if getattr(dp, '_pit_path_rounds', None) is None:
            host_fallbacks += 1
        if not np.array_equal(got2, want2, equal_nan=True):
            print('MISMATCH after pit_drain_paths', rec, int((got2 != want2).sum()), 'cells'); sys.exit(1)
        done += 1
    print('device conditioning soak ok: %d random tiles up to case %d in %.0f s (%d took the host loop for the paths, %d the host fill_flats)'
          % (done, k, time.time() - t0, host_fallbacks, globals().get('host_flats', 0)))


if __name__ == '__main__':
    main()
