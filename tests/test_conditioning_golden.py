"""Host-side elevation conditioning (pydem_amd/conditioning.py) against arrays captured from the unmodified
reference after each stage: calc_fill_pit_artifacts, calc_fill_flats, calc_pit_drain_paths
(dem_processing.py:396-579).  Bit-exact, CPU only (the conditioning is numpy/scipy code)."""
import warnings

import numpy as np
import pytest

from conftest import golden_names, load_golden
from pydem_amd import conditioning
import conditioning_numpy


def _cases():
    return [n for n in golden_names() + golden_names('g7_') if 'elev_drained' in load_golden(n) or 'elev_filled' in load_golden(n)]


@pytest.mark.parametrize('name', _cases())
def test_conditioning_matches_reference(name):
    g = load_golden(name)
    kw = g['kwargs']
    elev = g['in_elev'].copy()
    sea = kw.get('fill_flats_below_sea', False)
    if kw.get('fill_flats', True):
        if kw.get('maximum_pit_area', 32.0):
            art = conditioning.fill_pit_artifacts(elev, kw.get('maximum_pit_area', 32.0), sea)
            assert art.dtype == g['elev_artifacts'].dtype
            assert np.array_equal(art, g['elev_artifacts'], equal_nan=True)
        filled = conditioning.fill_flats(elev, kw.get('maximum_pit_area', 32.0), sea, kw.get('fill_flats_source_tol', 1),
                                         kw.get('fill_flats_peaks', True), kw.get('fill_flats_pits', True))
        assert np.array_equal(filled, g['elev_filled'], equal_nan=True)
        elev = filled
    if kw.get('drain_pits_path', True):
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            out, _, _ = conditioning.pit_drain_paths(np.array(elev), g['in_dX'], g['in_dY'], kw.get('drain_pits_max_iter', 300),
                                                     kw.get('drain_pits_max_dist', 32), kw.get('drain_pits_max_dist_XY', None), sea)
        assert out.dtype == g['elev_drained'].dtype
        assert np.array_equal(out, g['elev_drained'], equal_nan=True)


@pytest.mark.parametrize('kind,shape,seed', [('srtm', (160, 200), 5), ('srtm', (257, 129), 6), ('quant', (192, 192), 7),
                                            ('fractal', (128, 160), 8), ('f32', (150, 170), 9)])
def test_native_loops_match_numpy_loops(kind, shape, seed):
    """csrc/cond_host.cpp against the numpy versions it was written from, on tiles with thousands of flats,
    quantisation pits, plateaus on the tile edge and summit plateaus (bit for bit, all three stages)."""
    from pydem_amd import synth
    n, m = shape
    if kind == 'srtm':
        elev = synth.srtm_int16(n, m, seed=seed)
    elif kind == 'quant':
        elev = np.rint(synth.fractal(n, m, seed=seed, top_shift=6, n_octaves=6, zrange=40.0)).astype('int32')
    elif kind == 'f32':
        elev = (np.rint(synth.fractal(n, m, seed=seed, top_shift=6, n_octaves=6, zrange=60.0) * 2) / 2).astype('float32')
    else:
        elev = synth.fractal(n, m, seed=seed, top_shift=6, n_octaves=7)
    dX = 25.0 + 0.01 * np.arange(n - 1)
    dY = 31.0 - 0.004 * np.arange(n - 1)
    a1 = conditioning.fill_pit_artifacts(elev)
    a0 = conditioning_numpy.fill_pit_artifacts(elev)
    assert a1.dtype == a0.dtype and np.array_equal(a1, a0)
    f1 = conditioning.fill_flats(elev)
    f0 = conditioning_numpy.fill_flats(elev)
    assert np.array_equal(f1, f0, equal_nan=True)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        p1, bad1, it1 = conditioning.pit_drain_paths(f1.copy(), dX, dY)
        p0, bad0, it0 = conditioning_numpy.pit_drain_paths(f0.copy(), dX, dY)
    assert np.array_equal(p1, p0, equal_nan=True) and bad1 == bad0 and it1 == it0
    # options that take other branches: no peaks / no pits, tolerance 0, distance limits
    g1 = conditioning.fill_flats(elev, fill_flats_source_tol=0, fill_flats_peaks=False, fill_flats_pits=False)
    g0 = conditioning_numpy.fill_flats(elev, fill_flats_source_tol=0, fill_flats_peaks=False, fill_flats_pits=False)
    assert np.array_equal(g1, g0, equal_nan=True)
    # pit paths on the raw surface: integer surfaces truncate the path values, float32 surfaces round them (:539)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        r1 = conditioning.pit_drain_paths(elev.copy(), dX, dY)
        r0 = conditioning_numpy.pit_drain_paths(elev.copy(), dX, dY)
    assert r1[0].dtype == r0[0].dtype == elev.dtype and np.array_equal(r1[0], r0[0], equal_nan=True) and r1[1:] == r0[1:]
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        q1 = conditioning.pit_drain_paths(f1.copy(), dX, dY, drain_pits_max_iter=20, drain_pits_max_dist=6, drain_pits_max_dist_XY=150.0)
        q0 = conditioning_numpy.pit_drain_paths(f0.copy(), dX, dY, drain_pits_max_iter=20, drain_pits_max_dist=6, drain_pits_max_dist_XY=150.0)
    assert np.array_equal(q1[0], q0[0], equal_nan=True) and q1[1:] == q0[1:]
