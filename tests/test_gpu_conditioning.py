"""Elevation conditioning on the device (csrc/cond_device.hip through DEMProcessor.calc_fill_pit_artifacts /
calc_fill_flats) against (a) the arrays captured from the unmodified reference after each stage (g5_* / g7_*
goldens) and (b) the host implementation (pydem_amd/conditioning.py, itself pinned by those goldens) on seeded
random tiles: all dtypes, plateaus, sea level, every option.  Bit for bit."""
import warnings

import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _cases():
    return [n for n in golden_names() + golden_names('g7_') if 'elev_filled' in load_golden(n)]


def _dp(elev, **kw):
    from pydem_amd import DEMProcessor
    return DEMProcessor(elev=elev, dX=30.0, dY=30.0, **kw)


@pytest.mark.parametrize('name', _cases())
def test_device_conditioning_matches_reference(name):
    g = load_golden(name)
    kw = g['kwargs']
    if not kw.get('fill_flats', True):
        pytest.skip("no flats step in this golden")
    opts = {k: kw[k] for k in ('maximum_pit_area', 'fill_flats_below_sea', 'fill_flats_source_tol', 'fill_flats_peaks', 'fill_flats_pits')
            if k in kw}
    if kw.get('maximum_pit_area', 32.0) and 'elev_artifacts' in g:
        dp = _dp(g['in_elev'].copy(), **opts)
        dp.calc_fill_pit_artifacts()
        assert np.asarray(dp.elev).dtype == g['elev_artifacts'].dtype
        assert np.array_equal(dp.elev, g['elev_artifacts'], equal_nan=True)
    dp = _dp(g['in_elev'].copy(), **opts)
    dp.calc_fill_flats()
    assert 'elev' in dp._on_device              # (tiles with no-data cells too: the masks replay scipy's filter, csrc/cond_device.hip)
    assert np.asarray(dp.elev).dtype == np.float64
    assert np.array_equal(dp.elev, g['elev_filled'], equal_nan=True)


def _random_tile(k):
    from pydem_amd import synth
    rng = np.random.default_rng(7000 + k)
    n, m = int(rng.integers(3, 220)), int(rng.integers(3, 220))
    ts = int(rng.integers(2, 7))
    z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=ts, n_octaves=int(rng.integers(2, ts + 1)),
                      zmin=float(rng.choice([1.0, -10.0])), zrange=float(rng.choice([300.0, 40.0, 9.0])))
    kind = rng.choice(['int16', 'int32', 'quant', 'f64', 'f32', 'lake'])
    if rng.random() < 0.3:
        z[z < 0] = 0.0
    if kind == 'int16':
        z = np.rint(z).astype(np.int16)
    elif kind == 'int32':
        z = np.rint(z).astype(np.int32)
    elif kind == 'quant':
        z = np.rint(z)
    elif kind == 'f32':
        z = np.rint(z * 2).astype(np.float32) / np.float32(2)
    elif kind == 'lake':
        z = np.maximum(np.rint(z), np.quantile(z, 0.6)).astype(np.int16)          # one big exact plateau
    opt = dict(maximum_pit_area=float(rng.choice([32.0, 4.0, 0.0])), fill_flats_below_sea=bool(rng.random() < 0.3),
               fill_flats_source_tol=int(rng.choice([1, 0, 3])), fill_flats_peaks=bool(rng.random() < 0.7),
               fill_flats_pits=bool(rng.random() < 0.7))
    return z, opt


@pytest.mark.parametrize('block', range(4))
def test_device_conditioning_matches_host_twin(block):
    from pydem_amd import conditioning as C
    for k in range(block * 40, block * 40 + 40):
        z, opt = _random_tile(k)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            want_a = C.fill_pit_artifacts(z.copy(), opt['maximum_pit_area'] or 32.0, opt['fill_flats_below_sea'])
            want_f = C.fill_flats(z.copy(), **opt)
            dp = _dp(z.copy(), **opt)
            dp.maximum_pit_area = opt['maximum_pit_area'] or 32.0
            dp.calc_fill_pit_artifacts()
            got_a = np.asarray(dp.elev)
            dp = _dp(z.copy(), **opt)
            dp.calc_fill_flats()
            got_f = np.asarray(dp.elev)
        assert got_a.dtype == np.asarray(want_a).dtype, (k, z.dtype)
        assert np.array_equal(got_a, want_a), "case %d (%s, %s): artefact step differs on %d cells" % (k, z.shape, z.dtype, int((got_a != want_a).sum()))
        assert np.array_equal(got_f, want_f, equal_nan=True), \
            "case %d (%s, %s, %r): fill_flats differs on %d cells" % (k, z.shape, z.dtype, opt, int((got_f != want_f).sum()))


def test_large_plateau_tile_matches_host_twin():
    """BASELINE.json config 5 terrain at a size the host finishes in seconds: int16 with lakes (big exact plateaus)."""
    from pydem_amd import conditioning as C, synth
    z = synth.srtm_int16(1100, 900, seed=3)
    want = C.fill_flats(z.copy())
    dp = _dp(z.copy())
    dp.calc_fill_flats()
    assert 'elev' in dp._on_device
    assert np.array_equal(dp.elev, want)


# ---- pit drain paths (csrc/cond_paths.hip) ------------------------------------------------------------------
def _drained_cases():
    return [n for n in golden_names() + golden_names('g7_') if 'elev_drained' in load_golden(n)]


@pytest.mark.parametrize('name', _drained_cases())
def test_device_pit_paths_match_reference(name):
    from pydem_amd import DEMProcessor
    g = load_golden(name)
    kw = g['kwargs']
    opts = {k: kw[k] for k in ('maximum_pit_area', 'fill_flats_below_sea', 'fill_flats_source_tol', 'fill_flats_peaks', 'fill_flats_pits',
                               'drain_pits_max_iter', 'drain_pits_max_dist', 'drain_pits_max_dist_XY') if k in kw}
    dp = DEMProcessor(elev=g['in_elev'].copy(), dX=g['in_dX'], dY=g['in_dY'], **opts)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        if kw.get('fill_flats', True):
            dp.calc_fill_flats()
        dp.calc_pit_drain_paths()
    got = np.asarray(dp.elev)
    assert got.dtype == g['elev_drained'].dtype
    assert np.array_equal(got, g['elev_drained'], equal_nan=True)
    if name.startswith('g7_nan'):               # no-data tiles stay on the device (round 4): neither step took the host loops
        assert getattr(dp, '_pit_path_rounds', None) is not None


@pytest.mark.parametrize('block', range(3))
def test_device_conditioning_with_nodata_matches_host_twin(block):
    """Random tiles with no-data cells (blocks, margins, scattered voids, the sea as NaN: tools/soak_conditioning_device.py with
    SOAK_NAN=1) through the device conditioning against the host implementation, whose masks come from scipy itself: the
    replay of scipy's ring filter (csrc/cond_device.hip k_cond_ring_*), regions of mixed height, pits next to voids -- bit for
    bit after each step, on the device for both (reference: pydem/dem_processing.py:396-579 with utils.py:342-402)."""
    import os
    import sys
    from conftest import ROOT
    sys.path.insert(0, os.path.join(ROOT, 'tools'))
    old = os.environ.get('SOAK_NAN')
    os.environ['SOAK_NAN'] = '1'
    try:
        import soak_conditioning_device as S
        from pydem_amd import DEMProcessor, conditioning as C
        seen_nan = 0
        for k in range(block * 25, block * 25 + 25):
            rec, z, o, dX, dY = S.make_case(k)
            n = z.shape[0]
            with warnings.catch_warnings():
                warnings.simplefilter('ignore')
                want1 = C.fill_flats(z, o['maximum_pit_area'], o['fill_flats_below_sea'], o['fill_flats_source_tol'], o['fill_flats_peaks'], o['fill_flats_pits'])
                want2, _, _ = C.pit_drain_paths(want1.copy(), np.full(n - 1, dX), np.full(n - 1, dY), o['drain_pits_max_iter'], o['drain_pits_max_dist'],
                                                o['drain_pits_max_dist_XY'], o['fill_flats_below_sea'])
                dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, **o)
                dp.calc_fill_flats()
                assert 'elev' in dp._on_device, rec
                assert np.array_equal(np.array(dp.elev), want1, equal_nan=True), rec
                dp.calc_pit_drain_paths()
                assert np.array_equal(np.array(dp.elev), want2, equal_nan=True), rec
            seen_nan += int(np.isnan(z).any())
        assert seen_nan >= 15
    finally:
        if old is None:
            os.environ.pop('SOAK_NAN', None)
        else:
            os.environ['SOAK_NAN'] = old


@pytest.mark.parametrize('block', range(4))
def test_device_pit_paths_match_host_twin(block):
    """Random tiles (after the flats step, like the reference's pipeline, and raw float64 surfaces with plateaus left in):
    the order-preserving parallel schedule against the sequential host loop, bit for bit; the fallback to the host loop
    (a later pit's path met by an earlier pit's regrowth) must stay rare."""
    from pydem_amd import DEMProcessor, conditioning as C
    fell_back = 0
    for k in range(block * 30, block * 30 + 30):
        z, opt = _random_tile(k)
        rng = np.random.default_rng(k)
        n = z.shape[0]
        dX = 25.0 + 0.01 * np.arange(n - 1); dY = 31.0 - 0.004 * np.arange(n - 1)
        popt = dict(drain_pits_max_iter=int(rng.choice([300, 300, 6])), drain_pits_max_dist=int(rng.choice([32, 32, 3])),
                    drain_pits_max_dist_XY=(float(rng.uniform(30, 200)) if rng.random() < 0.2 else None))
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            if rng.random() < 0.7:
                base = C.fill_flats(z.copy(), **opt)
            else:
                base = np.ascontiguousarray(z, np.float64)
            if np.isnan(base).any():
                continue
            want, bad, used = C.pit_drain_paths(base.copy(), dX, dY, fill_flats_below_sea=opt['fill_flats_below_sea'], **popt)
            dp = DEMProcessor(elev=base.copy(), dX=dX, dY=dY, fill_flats_below_sea=opt['fill_flats_below_sea'], **popt)
            res = dp._pit_paths_on_device()
            if res is None:
                fell_back += 1
                dp.calc_pit_drain_paths()
            got = np.asarray(dp.elev)
        assert np.array_equal(got, want), "case %d (%s): paths differ on %d cells" % (k, z.shape, int((got != want).sum()))
        if res is not None:
            assert res[0] == bad and res[1] == used, (k, res, bad, used)
    assert fell_back <= 3, "%d of 30 tiles fell back to the host loop" % fell_back


def test_device_pit_paths_plateau_tile():
    """SRTM-like int16 tile with lakes: after the flats step the lake centres are pits whose region grows for all 300
    iterations (large-window simulations)."""
    from pydem_amd import DEMProcessor, conditioning as C, synth
    z = synth.srtm_int16(900, 1100, seed=3)
    dX = np.full(899, 30.0); dY = np.full(899, 30.0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        base = C.fill_flats(z.copy())
        want, bad, used = C.pit_drain_paths(base.copy(), dX, dY)
        dp = DEMProcessor(elev=base.copy(), dX=dX, dY=dY)
        res = dp._pit_paths_on_device()
    assert res is not None, "fell back to the host loop"
    assert np.array_equal(np.asarray(dp.elev), want)
    assert (res[0], res[1]) == (bad, used)


@pytest.mark.parametrize('dtype', ['int16', 'int32', 'float32'])
def test_device_pit_paths_keep_the_arrays_dtype(dtype):
    """calc_pit_drain_paths on an integer / float32 surface (fill_flats off): the reference edits the array in ITS dtype
    (dem_processing.py:535-539: float32 difference, path values truncated / rounded on assignment) and sorts the pits on keys of
    that dtype (:450).  The device path does the same (dtype modes of csrc/cond_paths.hip) and must equal the sequential host
    loop, which the reference's int16 / float32 goldens pin, on random tiles -- without falling back to it."""
    from pydem_amd import DEMProcessor, conditioning as C, synth
    done = 0
    for k in range(12):
        rng = np.random.default_rng(500 + k)
        n, m = int(rng.integers(40, 300)), int(rng.integers(40, 300))
        z = synth.fractal(n, m, seed=int(rng.integers(0, 1 << 30)), top_shift=5, n_octaves=4, zmin=1.0, zrange=float(rng.choice([40.0, 400.0])))
        z = (np.rint(z) if dtype != 'float32' else z).astype(dtype)
        dX = np.full(n - 1, 30.0); dY = np.full(n - 1, 25.0)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            want, bad, used = C.pit_drain_paths(z.copy(), dX, dY)
            dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, fill_flats=False)
            res = dp._pit_paths_on_device()
        if res is None:
            continue                                   # (the speculative schedule may give up: rare, counted below)
        got = np.asarray(dp.elev)
        assert got.dtype == np.dtype(dtype) and want.dtype == np.dtype(dtype)
        assert np.array_equal(got, want), "case %d: %d cells differ" % (k, int((got != want).sum()))
        assert (res[0], res[1]) == (bad, used)
        done += 1
    assert done >= 10


def test_device_pit_paths_trail_capacity_boundary():
    """Flat-floored basins whose flood holds 504 .. 529 cells when it meets its outlet -- around the 512 trail entries of the
    small-window simulation (the chain is pruned in place at the END of the trail, one entry past it): 512 fits exactly, 513 goes
    on to the medium window (trail in global memory, pruned in place as well).  Each basin: a pit one unit below the floor in the
    middle, a sink hole five units below it diagonally outside the corner the rings reach last (itself a pit that fails)."""
    from pydem_amd import DEMProcessor, conditioning as C
    n = m = 420
    ii, jj = np.meshgrid(np.arange(n, dtype=np.float64), np.arange(m, dtype=np.float64), indexing='ij')
    z = 5000.0 + 0.5 * ii + 0.3 * jj                                  # a tilted plane: no pits of its own
    sizes = [(21, 24), (22, 23), (15, 34), (16, 32), (19, 27), (23, 23)]
    expect = []
    for q, (rows, cols) in enumerate(sizes):
        r0, c0 = 40 + 130 * (q // 3), 40 + 130 * (q % 3)
        f = 1000.0 + 50.0 * q
        z[r0:r0 + rows, c0:c0 + cols] = f
        z[r0 + rows // 2, c0 + cols // 2] = f - 1.0
        z[r0 - 1, c0 - 1] = f - 5.0
        expect.append((r0 + rows // 2, c0 + cols // 2, r0 - 1, c0 - 1, rows * cols))
    dX = np.full(n - 1, 30.0); dY = np.full(n - 1, 30.0)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        want, bad, used = C.pit_drain_paths(z.copy(), dX, dY)
        dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY)
        res = dp._pit_paths_on_device()
    assert res is not None, "fell back to the host loop"
    got = np.asarray(dp.elev)
    assert np.array_equal(got, want), "paths differ on %d cells" % int((got != want).sum())
    assert (res[0], res[1]) == (bad, used)
    for pi, pj, si, sj, cells in expect:                               # every basin's pit got its path to the sink hole
        changed = (want != z)[pi - 40:pi + 40, pj - 40:pj + 40].sum()
        assert changed >= 8, (cells, changed)
        assert want[si, sj] == z[si, sj]
