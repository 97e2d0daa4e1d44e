#!/usr/bin/env python
"""Differential soak of the conditioning step (pydem_amd/conditioning.py: fill_pit_artifacts, fill_flats,
pit_drain_paths) against the unmodified reference on random small tiles with random options; every stage bit for
bit (dtype included).  Build container only.   bash run.sh soak_conditioning_reference.py [seconds] [first_case]"""
import os
import sys
import time
import warnings

import gen_golden as G     # noqa: F401
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from pydem_amd import conditioning, synth   # noqa: E402


def make_case(k):
    rng = np.random.default_rng(61000 + k)
    n, m = int(rng.integers(3, 48)), int(rng.integers(3, 48))
    ts = int(rng.integers(2, 6))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -10.0])), zrange=float(rng.choice([300.0, 40.0, 9.0])))
    kind = rng.choice(['int16', 'quant', 'f64', 'f32'])
    if rng.random() < 0.3:
        z[z < 0] = 0.0
    if kind == 'int16':
        z = np.rint(z).astype(np.int16)
    elif kind == 'quant':
        z = np.rint(z)
    elif kind == 'f32':
        z = z.astype(np.float32)
    if z.dtype.kind == 'f' and rng.random() < 0.2:
        i0, j0 = int(rng.integers(0, n)), int(rng.integers(0, m))
        z[i0:i0 + int(rng.integers(1, 8)), j0:j0 + int(rng.integers(1, 8))] = np.nan
    opt = {}
    if rng.random() < 0.25: opt['fill_flats'] = False
    if rng.random() < 0.2: opt['drain_pits_path'] = False
    if rng.random() < 0.3: opt['maximum_pit_area'] = float(rng.choice([4.0, 0.0]))
    if rng.random() < 0.3: opt['fill_flats_below_sea'] = True
    if rng.random() < 0.3: opt['fill_flats_source_tol'] = int(rng.choice([0, 3]))
    if rng.random() < 0.3: opt['fill_flats_peaks'] = False
    if rng.random() < 0.3: opt['fill_flats_pits'] = False
    if rng.random() < 0.2: opt['drain_pits_max_iter'] = int(rng.integers(2, 30))
    if rng.random() < 0.2: opt['drain_pits_max_dist'] = int(rng.integers(1, 10))
    if rng.random() < 0.1: opt['drain_pits_max_dist_XY'] = float(rng.uniform(30, 200))
    opt['drain_pits'] = False          # the rest of the path is soaked elsewhere
    return dict(case=k, shape=(n, m), dtype=str(z.dtype), options=opt), z, opt


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); done = 0; skipped = 0
    warnings.simplefilter('ignore')
    devnull = open(os.devnull, 'w')
    while time.time() - t0 < budget:
        rec, z, kw = make_case(k)
        k += 1
        out, sys.stdout = sys.stdout, devnull
        try:
            try:
                g = G.run_case(z, 30.0, 25.0, **kw)
            except Exception:
                skipped += 1           # inputs the reference itself cannot process
                continue
        finally:
            sys.stdout = out
        elev = z.copy()
        sea = kw.get('fill_flats_below_sea', False)
        errs = []
        if kw.get('fill_flats', True):
            if kw.get('maximum_pit_area', 32.0):
                art = conditioning.fill_pit_artifacts(elev, kw.get('maximum_pit_area', 32.0), sea)
                if art.dtype != g['elev_artifacts'].dtype or not np.array_equal(art, g['elev_artifacts'], equal_nan=(art.dtype.kind == 'f')):
                    errs.append('artifacts')
            filled = conditioning.fill_flats(elev, kw.get('maximum_pit_area', 32.0), sea, kw.get('fill_flats_source_tol', 1),
                                             kw.get('fill_flats_peaks', True), kw.get('fill_flats_pits', True))
            if not np.array_equal(filled, g['elev_filled'], equal_nan=True):
                errs.append('fill_flats')
            elev = filled
        if kw.get('drain_pits_path', True) and not errs:
            o, _, _ = conditioning.pit_drain_paths(np.array(elev), g['in_dX'], g['in_dY'], kw.get('drain_pits_max_iter', 300),
                                                   kw.get('drain_pits_max_dist', 32), kw.get('drain_pits_max_dist_XY', None), sea)
            if o.dtype != g['elev_drained'].dtype or not np.array_equal(o, g['elev_drained'], equal_nan=(o.dtype.kind == 'f')):
                errs.append('pit_drain_paths')
        if errs:
            print('MISMATCH', rec, errs)
            sys.exit(1)
        done += 1
    print('conditioning reference soak ok: %d random tiles (%d skipped) up to case %d in %.0f s' % (done, skipped, k, time.time() - t0))


if __name__ == '__main__':
    main()
