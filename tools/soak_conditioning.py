#!/usr/bin/env python
"""Randomised CPU soak of the conditioning step: the native per-region / per-pit loops (libpydem_hip.so, host code)
against the numpy restatements they were written from (tests/conditioning_numpy.py, themselves pinned by the
reference goldens g7_*): random int16 / float tiles with plateaus, sea, nodata and random options; exact equality.
soak_conditioning.py [seconds] [first_case]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests'))
from pydem_amd import conditioning as C, synth   # noqa: E402
import conditioning_numpy as CN                  # noqa: E402


def make_case(k):
    rng = np.random.default_rng(9000 + k)
    n, m = int(rng.integers(3, 160)), int(rng.integers(3, 160))
    ts = int(rng.integers(2, 7))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -10.0])), zrange=float(rng.choice([300.0, 40.0, 9.0])))
    kind = rng.choice(['int16', 'quant', 'f64'])
    if rng.random() < 0.3:
        z[z < 0] = 0.0
    if kind == 'int16':
        z = np.rint(z).astype(np.int16)
    elif kind == 'quant':
        z = np.rint(z)
    if z.dtype.kind == 'f' and rng.random() < 0.25:
        i0, j0 = int(rng.integers(0, n)), int(rng.integers(0, m))
        z[i0:i0 + int(rng.integers(1, 12)), j0:j0 + int(rng.integers(1, 12))] = np.nan
    opt = dict(maximum_pit_area=float(rng.choice([32.0, 4.0, 0.0])), fill_flats_below_sea=bool(rng.random() < 0.3),
               fill_flats_source_tol=int(rng.choice([1, 0, 3])), fill_flats_peaks=bool(rng.random() < 0.7),
               fill_flats_pits=bool(rng.random() < 0.7))
    popt = dict(drain_pits_max_iter=int(rng.choice([300, 6])), drain_pits_max_dist=int(rng.choice([32, 3])),
                drain_pits_max_dist_XY=(float(rng.uniform(30, 200)) if rng.random() < 0.2 else None),
                fill_flats_below_sea=opt['fill_flats_below_sea'])
    return dict(case=k, shape=(n, m), dtype=str(z.dtype), options=opt, path_options=popt), z, opt, popt


def same(a, b):
    return a.dtype == b.dtype and np.array_equal(a, b, equal_nan=(a.dtype.kind == 'f'))


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); done = 0
    warnings.simplefilter('ignore')
    while time.time() - t0 < budget:
        rec, z, opt, popt = make_case(k)
        k += 1
        a = C.fill_pit_artifacts(z.copy(), opt['maximum_pit_area'], opt['fill_flats_below_sea'])
        b = CN.fill_pit_artifacts(z.copy(), opt['maximum_pit_area'], opt['fill_flats_below_sea'])
        if not same(np.asarray(a), np.asarray(b)):
            print('MISMATCH fill_pit_artifacts', rec); sys.exit(1)
        a = C.fill_flats(z.copy(), **opt); b = CN.fill_flats(z.copy(), **opt)
        if not same(a, b):
            print('MISMATCH fill_flats', rec, int((a != b).sum())); sys.exit(1)
        n = z.shape[0]
        dX, dY = 30.0 * np.ones(n - 1), 25.0 + 0.01 * np.arange(n - 1)
        pa = C.pit_drain_paths(a.copy(), dX, dY, **popt); pb = CN.pit_drain_paths(b.copy(), dX, dY, **popt)
        if not same(pa[0], pb[0]) or pa[1:] != tuple(pb[1:]):
            print('MISMATCH pit_drain_paths', rec, pa[1:], pb[1:]); sys.exit(1)
        done += 1
    print('conditioning soak ok: %d random cases up to %d in %.0f s' % (done, k, time.time() - t0))


if __name__ == '__main__':
    main()
