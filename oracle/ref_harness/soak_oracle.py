#!/usr/bin/env python
"""Differential soak of the CPU oracle against the UNMODIFIED reference (imported through load_reference) on random
small tiles: every array bit for bit, pit triplets as sorted sets.  Runs in the build container only (the reference is
not on the GPU box).   bash run.sh soak_oracle.py [seconds] [first_case]"""
import os
import sys
import time
import warnings

import gen_golden as G     # noqa: F401  (imports the reference with numpy on glibc libm)
import numpy as np

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO)
from oracle import oracle as O          # noqa: E402
from pydem_amd import synth             # noqa: E402


def make_case(k):
    rng = np.random.default_rng(70000 + k)
    n, m = int(rng.integers(3, 56)), int(rng.integers(3, 56))
    ts = int(rng.integers(2, 6))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -15.0, -1.5])), zrange=float(rng.choice([500.0, 60.0, 12.0])))
    mode = rng.choice(['f64', 'quant', 'f32', 'int16'], p=[0.4, 0.3, 0.15, 0.15])
    if rng.random() < 0.3:
        z[z < 0] = 0.0
    if mode == 'quant':
        z = np.rint(z)
    elif mode == 'f32':
        z = z.astype(np.float32)
    elif mode == 'int16':
        z = np.rint(z).astype(np.int16)
    if z.dtype.kind == 'f' and rng.random() < 0.3:
        i0, j0 = int(rng.integers(0, n)), int(rng.integers(0, m))
        z[i0:i0 + int(rng.integers(1, 8)), j0:j0 + int(rng.integers(1, 8))] = np.nan
    if rng.random() < 0.5:
        dX, dY = float(rng.choice([30.0, 1.0, 12.5])), float(rng.choice([30.0, 1.0, 17.0]))
    else:
        a, b = float(rng.uniform(5, 40)), float(rng.uniform(5, 40))
        dX, dY = a + 0.3 * a / n * np.arange(n - 1), b - 0.2 * b / n * np.arange(n - 1)
    opt = dict(fill_flats=False, drain_pits_path=False)
    if rng.random() < 0.2: opt['drain_pits'] = False
    if rng.random() < 0.15: opt['drain_pits_min_border'] = True
    if rng.random() < 0.15: opt['drain_pits_max_iter'] = int(rng.integers(1, 40))
    if rng.random() < 0.15: opt['drain_pits_max_dist'] = int(rng.integers(1, 12))
    if rng.random() < 0.1: opt['drain_pits_max_dist_XY'] = float(rng.uniform(20, 300))
    if rng.random() < 0.1: opt.update(apply_uca_limit_edges=True, uca_saturation_limit=float(rng.uniform(1, 8)))
    if rng.random() < 0.1: opt.update(apply_twi_limits=True, apply_twi_limits_on_uca=True, twi_min_slope=0.01)
    return dict(case=k, shape=(n, m), dtype=str(z.dtype), options=opt), z, dX, dY, opt


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); done = 0
    warnings.simplefilter('ignore')
    devnull = open(os.devnull, 'w')
    while time.time() - t0 < budget:
        rec, z, dX, dY, opt = make_case(k)
        k += 1
        out, sys.stdout = sys.stdout, devnull
        try:
            g = G.run_case(z, dX, dY, **opt)
        finally:
            sys.stdout = out
        okw = {kk: v for kk, v in opt.items() if kk not in ('fill_flats', 'drain_pits_path')}
        o = O.OracleDEM(g['elev_final'], dX=g['in_dX'], dY=g['in_dY'], dX2=g['in_dX2'], dY2=g['in_dY2'], **okw)
        o.calc_twi()
        errs = []
        for nm, a, b in (('mag', o.mag, g['mag_final']), ('direction', o.direction, g['direction']), ('uca', o.uca, g['uca']),
                         ('proportion', o.proportion, g['proportion']), ('twi', o.calc_twi(), g['twi_ret'])):
            if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True):
                errs.append(nm)
        if not np.array_equal(o.flats.astype(bool), g['flats_final']): errs.append('flats')
        if not np.array_equal(o.section, g['section']): errs.append('section')
        if not np.array_equal(o.edge_todo, g['edge_todo']): errs.append('edge_todo')
        if not np.array_equal(o.edge_done, g['edge_done']): errs.append('edge_done')
        if 'pit_i' in g and opt.get('drain_pits', True):
            ref = sorted(zip(g['pit_i'].tolist(), g['pit_j'].tolist(), g['pit_prop'].tolist()))
            mine = sorted(zip(o.pit_i.tolist(), o.pit_j.tolist(), o.pit_prop.tolist()))
            if [r[:2] for r in ref] != [x[:2] for x in mine]: errs.append('pit assignments')
            elif not np.array_equal([r[2] for r in ref], [x[2] for x in mine], equal_nan=True): errs.append('pit weights')
        if errs:
            print('MISMATCH', rec, errs)
            sys.exit(1)
        done += 1
    print('oracle soak ok: %d random tiles bit-identical to the reference (up to case %d) in %.0f s' % (done, k, time.time() - t0))


if __name__ == '__main__':
    main()
