#!/usr/bin/env python
"""Randomised parity soak: device path against the CPU oracle on random tiles (shape, relief, quantisation, nodata,
sea level, dtype, spacing, pit options).  Runs for a number of seconds or cases and stops at the first mismatch,
printing the recipe that reproduces it.   soak_parity.py [seconds] [first_case]"""
import os
import sys
import time
import warnings

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O            # noqa: E402  (checker only)
from pydem_amd import DEMProcessor, synth  # noqa: E402

RTOL, ATOL = 1e-9, 1e-12
BIG = os.environ.get('SOAK_BIG') == '1'


def close(a, b, what):
    a = np.asarray(a, float); b = np.asarray(b, float)
    if not np.array_equal(np.isnan(a), np.isnan(b)):
        return "%s: NaN pattern differs" % what
    ok = np.isclose(a, b, rtol=RTOL, atol=ATOL, equal_nan=True)
    if not ok.all():
        return "%s: %d cells differ, worst rel %g" % (what, (~ok).sum(), np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))
    return None


def make_case(k):
    rng = np.random.default_rng(1000 + k)
    n, m = int(rng.integers(3, 700)), int(rng.integers(3, 700))
    if rng.random() < 0.15:
        n, m = int(rng.integers(3, 40)), int(rng.integers(3, 40))
    if rng.random() < 0.06:                                # slivers: one dimension minimal, the other long
        a, b = int(rng.integers(3, 8)), int(rng.integers(500, 20000))
        n, m = (a, b) if rng.random() < 0.5 else (b, a)
    if BIG:                                                # large tiles: plateau pits that need the big pit tiers, deep sweeps
        n, m = int(rng.integers(900, 2600)), int(rng.integers(900, 2600))
    rec = dict(case=k, shape=(n, m))
    ts = int(rng.integers(2, 8))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -20.0, -1.5])), zrange=float(rng.choice([1000.0, 200.0, 30.0, 8.0])))
    mode = rng.choice(['f64', 'quant', 'f32', 'int16', 'int32'], p=[0.35, 0.25, 0.15, 0.15, 0.10])
    if rng.random() < 0.3:
        z[z < 0] = 0.0                                     # sea level
        rec['sea'] = True
    if mode == 'quant':
        z = np.rint(z)
    elif mode == 'f32':
        z = z.astype(np.float32)
    elif mode == 'int16':
        z = np.rint(z).astype(np.int16)
    elif mode == 'int32':
        z = np.rint(z * 50).astype(np.int32)
    if z.dtype.kind == 'f' and rng.random() < 0.3:
        for _ in range(int(rng.integers(1, 4))):           # nodata blocks and specks
            i0, j0 = int(rng.integers(0, n)), int(rng.integers(0, m))
            z[i0:i0 + int(rng.integers(1, 20)), j0:j0 + int(rng.integers(1, 20))] = np.nan
        for _ in range(int(rng.integers(0, 10))):
            z[int(rng.integers(0, n)), int(rng.integers(0, m))] = np.nan
        rec['nodata'] = True
    rec['dtype'] = str(z.dtype)
    if rng.random() < 0.5:
        kw = dict(dX=float(rng.choice([30.0, 1.0, 12.5])), dY=float(rng.choice([30.0, 1.0, 17.0])))
    else:
        a, b = float(rng.uniform(5, 40)), float(rng.uniform(5, 40))
        gx, gy = 0.3 * a / n, 0.2 * b / n                   # spacings stay positive on tiles of any height
        kw = dict(dX=a + gx * np.arange(n - 1), dY=b - gy * np.arange(n - 1), dX2=a + gx * np.arange(n), dY2=b - gy * np.arange(n))
        rec['spacing'] = 'varying'
    opt = {}
    if rng.random() < 0.2: opt['drain_pits'] = False
    if rng.random() < 0.15: opt['drain_pits_min_border'] = True
    if rng.random() < 0.15: opt['drain_pits_max_iter'] = int(rng.integers(1, 40))
    if rng.random() < 0.15: opt['drain_pits_max_dist'] = int(rng.integers(1, 12))
    if rng.random() < 0.1: opt['drain_pits_max_dist_XY'] = float(rng.uniform(20, 300))
    if rng.random() < 0.1: opt.update(apply_uca_limit_edges=True, uca_saturation_limit=float(rng.uniform(1, 8)))
    if rng.random() < 0.1: opt.update(apply_twi_limits=True, apply_twi_limits_on_uca=True, twi_min_slope=0.01)
    rec['options'] = opt
    return rec, z, kw, opt


def run_case(k):
    rec, z, kw, opt = make_case(k)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        o = O.OracleDEM(z, **kw, **opt)
        o.calc_twi()
        dp = DEMProcessor(elev=z, fill_flats=False, drain_pits_path=False, **kw, **opt)
        twi = dp.calc_twi()
    errs = []
    if not np.array_equal(dp.flats, o.flats.astype(bool)): errs.append('flats')
    if not np.array_equal(dp.section, o.section): errs.append('section')
    if not np.array_equal(dp.edge_todo, o.edge_todo): errs.append('edge_todo')
    if not np.array_equal(dp.edge_done, o.edge_done): errs.append('edge_done')
    for nm, a, b in (('mag', dp.mag, o.mag), ('direction', dp.direction, o.direction), ('proportion', dp.proportion, o.proportion),
                     ('uca', dp.uca, o.uca), ('twi', twi, o.twi / 10)):
        e = close(a, b, nm)
        if e: errs.append(e)
    if opt.get('drain_pits', True):
        src, dst, w = dp._tile.pit_edges()
        got = sorted(zip(src.tolist(), dst.tolist())); ref = sorted(zip(o.pit_i.tolist(), o.pit_j.tolist()))
        if got != ref: errs.append('pit assignments (%d vs %d edges)' % (len(got), len(ref)))
        if dp.timings['n_pits_undrained'] != o.n_warn: errs.append('undrained count')
    return rec, errs


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); done = 0
    while time.time() - t0 < budget:
        rec, errs = run_case(k)
        if errs:
            print('MISMATCH', rec, errs)
            sys.exit(1)
        k += 1; done += 1
    print('soak ok: %d random cases (first %d, next %d) in %.0f s' % (done, k - done, k, time.time() - t0))


if __name__ == '__main__':
    main()
