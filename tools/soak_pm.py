#!/usr/bin/env python
"""Randomised directory-flow soak: ProcessManager with the device DEMProcessor against the same ProcessManager with
the CPU-oracle processor (tests/oracle_processor.py) on random mosaics (raster, tile grid, overlap, nodata, sea level,
pit handling on/off).  Exercises the device edge-update rounds in the reference's visiting order, or -- SOAK_POOL=1 -- the
multi-worker schedule (deterministic waves, incremental rounds, device edge board) against the numpy strip rules with the
oracle processor at a random pool width.  Stops at the first mismatch.   soak_pm.py [seconds] [first_case]"""
import os
import shutil
import sys
import tempfile
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tests'))
from oracle_processor import OracleProcessor   # noqa: E402  (checker only)
from pydem_amd import _ffi, process_manager, synth    # noqa: E402
from pydem_amd.parallel import RcclTransport   # noqa: E402


_COMM = []


def comm():
    if not _COMM:
        _COMM.append(_ffi.Comm(1, 0, _ffi.Comm.unique_id(), 0))
    return _COMM[0]


def make_case(k):
    rng = np.random.default_rng(5000 + k)
    ny, nx = int(rng.integers(1, 4)), int(rng.integers(1, 4))
    ov = int(rng.integers(1, 4))
    scale = int(os.environ.get('SOAK_SCALE', '1'))            # larger tiles: longer cascades, wide frontiers, the level kernels
    n, m = int(rng.integers(12 * ny, 90 * ny + 1)) * scale, int(rng.integers(12 * nx, 90 * nx + 1)) * scale
    ts = int(rng.integers(2, 7))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -15.0])), zrange=float(rng.choice([500.0, 60.0, 12.0])))
    if rng.random() < 0.3:
        z[z < 0] = 0.0
    if rng.random() < 0.35:
        z = np.rint(z)
    if rng.random() < 0.3:
        i0, j0 = int(rng.integers(0, n)), int(rng.integers(0, m))
        z[i0:i0 + int(rng.integers(1, 15)), j0:j0 + int(rng.integers(1, 15))] = np.nan
    dkw = dict(drain_pits_path=False, fill_flats=False)
    if rng.random() < 0.25:
        dkw['drain_pits'] = False
    rng2 = np.random.default_rng(77000 + k)                   # TWI options on their own stream (earlier cases keep their rasters)
    if rng2.random() < 0.3:
        dkw.update(apply_twi_limits=bool(rng2.random() < 0.7), apply_twi_limits_on_uca=bool(rng2.random() < 0.7),
                   twi_min_slope=float(rng2.choice([0.01, 1e-3, 0.2])), uca_saturation_limit=float(rng2.choice([4.0, 32.0, 1.5])))
        if rng2.random() < 0.5:
            dkw['twi_min_area'] = float(rng2.choice([1.0, 25.0]))
    return dict(case=k, shape=(n, m), grid=(ny, nx), overlap=ov, options=dkw), z, ny, nx, ov, dkw


POOL = os.environ.get('SOAK_POOL') == '1'


def run(z, ny, nx, ov, dkw, cls, width=1):
    d = tempfile.mkdtemp()
    try:
        for t, (elev, bounds) in enumerate(synth.split_mosaic(z, ny, nx, ov)):
            np.savez(os.path.join(d, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
        process_manager.DEBUG = True
        kw = dict(tiles_in_flight=int(os.environ.get('SOAK_IN_FLIGHT', '1'))) if cls is None else dict(processor_cls=cls)
        if POOL:
            kw.update(n_workers=width, edge_mode='pool')
        pm = process_manager.ProcessManager(in_path=d, elev_conditioned=True, dem_proc_kwargs=dict(dkw), **kw)
        if cls is None and os.environ.get('SOAK_RCCL') == '1':      # strips through the RCCL transport (one rank)
            pm.transport = RcclTransport(pm, comm())
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pm.process_twi()
        return pm
    finally:
        process_manager.DEBUG = False
        shutil.rmtree(d, ignore_errors=True)


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    k = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    t0 = time.time(); done = 0; skipped = 0; cyclic = []
    while time.time() - t0 < budget:
        rec, z, ny, nx, ov, dkw = make_case(k)
        k += 1
        try:
            width = int(np.random.default_rng(77 + k).choice([2, 3, 8]))
            ref = run(z, ny, nx, ov, dkw, OracleProcessor, width)
        except Exception as e:                      # e.g. a degenerate tile grid the host logic rejects for both
            skipped += 1
            continue
        try:
            dev = run(z, ny, nx, ov, dkw, None, width)
        except RuntimeError as e:
            if 'circular drainage' in str(e):       # more unfinished cells than the sequential re-seed replay accepts
                cyclic.append(rec['case'])
                continue
            print('DEVICE RUN FAILED', rec, repr(e)[:300])
            sys.exit(1)
        errs = []
        if dev.edge_rounds != ref.edge_rounds or getattr(dev, 'edge_waves', 0) != getattr(ref, 'edge_waves', 0):
            errs.append('edge rounds %d vs %d, waves %d vs %d' % (dev.edge_rounds, ref.edge_rounds, getattr(dev, 'edge_waves', 0), getattr(ref, 'edge_waves', 0)))
        for i in range(ref.n_inputs):
            # where the edge corrections cancel a cell's area to (almost) exactly zero, the two summation orders leave
            # 0.0 on one side and +-1e-16 on the other: log() turns that into -inf vs NaN / -36; such cells are exempt
            # from the TWI comparison (their UCA is still compared, at atol 1e-12)
            zero_area = np.abs(np.asarray(ref.tile_result(i, 'uca_total'), float)) < 1e-9
            for key in ('aspect', 'slope', 'uca_total', 'twi'):
                a, b = np.asarray(dev.tile_result(i, key), float), np.asarray(ref.tile_result(i, key), float)
                if key == 'twi':
                    a = np.where(zero_area, 0.0, a); b = np.where(zero_area, 0.0, b)
                if not np.array_equal(np.isnan(a), np.isnan(b)) or not np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True):
                    errs.append('tile %d %s' % (i, key))
            for key in ('edge_todo', 'edge_done'):
                if not np.array_equal(dev.tile_result(i, key), ref.tile_result(i, key)):
                    errs.append('tile %d %s' % (i, key))
        if errs:
            print('MISMATCH', rec, errs[:8])
            sys.exit(1)
        done += 1
    print('pm soak ok: %d random mosaics (%d skipped, cyclic drainage refused on cases %s) up to case %d in %.0f s'
          % (done, skipped, cyclic, k, time.time() - t0))


if __name__ == '__main__':
    main()
