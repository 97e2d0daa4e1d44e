"""GPU parity: the HIP path (through the C-ABI, via pydem_amd.DEMProcessor) against
 (a) golden vectors captured from the unmodified reference (tests/golden/), and
 (b) the CPU oracle (oracle/pydem_oracle.c, itself bit-exact against those goldens) on seeded
     synthetic tiles.
Bars: bit-exact for flats / section / edge_todo / edge_done (integer and bool work);
      float64 mag, direction, proportion, uca, twi within RTOL below (device atan2/log are not
      glibc's, and the sweep adds in-edges in a different order) -- BASELINE.json asks for 1e-6.
"""
import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu

RTOL = 1e-11
ATOL = 1e-13


def _close(a, b, what):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, what
    assert np.array_equal(np.isnan(a), np.isnan(b)), "%s: NaN pattern differs" % what
    ok = np.isclose(a, b, rtol=RTOL, atol=ATOL, equal_nan=True)
    assert ok.all(), "%s: %d cells differ, worst rel %g" % (
        what, (~ok).sum(), np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


def _run_gpu(elev, g_or_o, **kw):
    from pydem_amd import DEMProcessor
    dp = DEMProcessor(elev=elev, fill_flats=False, drain_pits_path=False, **kw)
    return dp


def _golden_cases():
    out = []
    for name in golden_names():
        g = load_golden(name)
        kw = g['kwargs']
        if kw.get('drain_pits', True):
            continue          # pit->drain assignment on the device: see test_gpu_pits.py
        out.append(name)
    return out


@pytest.mark.parametrize('name', _golden_cases())
def test_hip_vs_reference_golden(name):
    g = load_golden(name)
    from pydem_amd import DEMProcessor
    dp = DEMProcessor(elev=g['elev_final'], dX=g['in_dX'], dY=g['in_dY'], dX2=g['in_dX2'], dY2=g['in_dY2'],
                      fill_flats=False, drain_pits_path=False, drain_pits=False)
    mag, direction = dp.calc_slopes_directions()
    _close(mag, g['mag'], 'mag')
    _close(direction, g['direction'], 'direction')
    assert np.array_equal(dp.flats, g['flats'])
    uca = dp.calc_uca()
    assert np.array_equal(dp.section, g['section'])
    _close(dp.proportion, g['proportion'], 'proportion')
    _close(uca, g['uca'], 'uca')
    assert np.array_equal(dp.edge_todo, g['edge_todo'])
    assert np.array_equal(dp.edge_done, g['edge_done'])
    twi = dp.calc_twi()
    _close(twi, g['twi_ret'], 'twi')
    _close(dp.twi, g['twi_attr'], 'twi attr')
    assert dp.twi_min_area == float(g['twi_min_area'])


@pytest.mark.parametrize('shape,seed,spacing', [((257, 383), 11, 'uniform'), ((512, 768), 12, 'varying'),
                                                ((1024, 1024), 13, 'uniform'), ((130, 67), 14, 'varying')])
def test_hip_vs_oracle_synthetic(shape, seed, spacing):
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth
    n, m = shape
    elev = synth.fractal(n, m, seed=seed, top_shift=7, n_octaves=7)
    if spacing == 'uniform':
        dX = dY = 30.0
        kw = dict(dX=dX, dY=dY)
    else:
        kw = dict(dX=25.0 + 0.01 * np.arange(n - 1), dY=31.0 - 0.004 * np.arange(n - 1),
                  dX2=25.0 + 0.01 * np.arange(n), dY2=31.0 - 0.004 * np.arange(n))
    o = O.OracleDEM(elev, drain_pits=False, **kw)
    o.calc_twi()
    dp = DEMProcessor(elev=elev, fill_flats=False, drain_pits_path=False, drain_pits=False, **kw)
    twi = dp.calc_twi()
    _close(dp.mag, o.mag, 'mag')
    _close(dp.direction, o.direction, 'direction')
    assert np.array_equal(dp.flats, o.flats.astype(bool))
    assert np.array_equal(dp.section, o.section)
    _close(dp.proportion, o.proportion, 'proportion')
    _close(dp.uca, o.uca, 'uca')
    assert np.array_equal(dp.edge_todo, o.edge_todo)
    assert np.array_equal(dp.edge_done, o.edge_done)
    _close(twi, o.twi / 10, 'twi')


def test_integer_dem_ties():
    """Quantised elevations: exact facet ties must resolve as in the reference ('first facet wins')."""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth
    elev = np.rint(synth.fractal(300, 260, seed=21, top_shift=6, n_octaves=6, zrange=40.0))
    o = O.OracleDEM(elev, dX=10.0, dY=10.0, drain_pits=False); o.calc_twi()
    dp = DEMProcessor(elev=elev.astype(np.int16), dX=10.0, dY=10.0, fill_flats=False, drain_pits_path=False,
                      drain_pits=False)
    dp.calc_twi()
    assert np.array_equal(dp.flats, o.flats.astype(bool))
    assert np.array_equal(dp.section, o.section)
    _close(dp.mag, o.mag, 'mag'); _close(dp.direction, o.direction, 'direction'); _close(dp.uca, o.uca, 'uca')


def test_synth_generator_bit_identical():
    from pydem_amd import _ffi, synth
    t = _ffi.Tile(100, 140)
    t.synth_fractal(seed=4, row0=5000, col0=77, n_octaves=8, top_shift=9, zmin=1.0, zrange=500.0)
    a = t.download(_ffi.ELEV)
    b = synth.fractal(100, 140, seed=4, row0=5000, col0=77, n_octaves=8, top_shift=9, zmin=1.0, zrange=500.0)
    assert np.array_equal(a, b)


def _default_cases():
    names = [n for n in golden_names() + golden_names('g7_')]
    return [n for n in names if load_golden(n)['kwargs'].get('fill_flats', True) or load_golden(n)['kwargs'].get('drain_pits_path', True)]


@pytest.mark.parametrize('name', _default_cases())
def test_full_pipeline_with_conditioning(name):
    """DEMProcessor with the reference's option set, from the RAW input: host conditioning
    (fill flats / pit drain paths) then the device path; every output against the reference."""
    g = load_golden(name)
    from pydem_amd import DEMProcessor
    import warnings
    dp = DEMProcessor(elev=g['in_elev'], dX=g['in_dX'], dY=g['in_dY'], dX2=g['in_dX2'], dY2=g['in_dY2'], **g['kwargs'])
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        twi = dp.calc_twi()
    assert np.array_equal(np.asarray(dp.elev, float), g['elev_final'].astype(float), equal_nan=True)     # (no-data goldens hold NaN)
    _close(dp.mag, g['mag_final'], 'mag')
    _close(dp.direction, g['direction'], 'direction')
    assert np.array_equal(dp.flats, g['flats_final'])
    assert np.array_equal(dp.section, g['section'])
    _close(dp.uca, g['uca'], 'uca')
    assert np.array_equal(dp.edge_todo, g['edge_todo'])
    assert np.array_equal(dp.edge_done, g['edge_done'])
    _close(twi, g['twi_ret'], 'twi')
    _close(dp.twi, g['twi_attr'], 'twi attr')


def test_geotiff_tile_through_elev_fn_constructor():
    """The reference's README entry point: DEMProcessor(elev_fn=<GeoTIFF>) (dem_processing.py:229-232).  The reference's
    own test raster goes through pydem_amd/raster.py into the device flow; with the spacing of the golden capture
    (the harness ran the reference with dX = dY = 1) every output must equal the reference's."""
    import os
    import warnings
    from pydem_amd import DEMProcessor
    g = load_golden('g3_tif32')
    fn = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_test_NN032_033_elev.tif')
    dp = DEMProcessor(elev_fn=fn, dX=g['in_dX'], dY=g['in_dY'], dX2=g['in_dX2'], dY2=g['in_dY2'], **g['kwargs'])
    assert len(dp.bounds) == 4 and len(dp.transform) >= 6
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        twi = dp.calc_twi()
    assert np.array_equal(np.asarray(dp.elev, float), g['elev_final'].astype(float))
    _close(dp.mag, g['mag_final'], 'mag')
    _close(dp.direction, g['direction'], 'direction')
    assert np.array_equal(dp.flats, g['flats_final']) and np.array_equal(dp.section, g['section'])
    _close(dp.uca, g['uca'], 'uca')
    assert np.array_equal(dp.edge_todo, g['edge_todo']) and np.array_equal(dp.edge_done, g['edge_done'])
    _close(twi, g['twi_ret'], 'twi')
    # ... and with the raster's own geodesic spacing (WGS-84 tile): the flow runs and every cell carries at least its own area
    dp2 = DEMProcessor(elev_fn=fn, fill_flats=False)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp2.calc_twi()
    assert dp2.dX.shape == (31,) and np.all(dp2.dX > 0)
    ok = ~np.isnan(dp2.uca)
    assert ok.any() and np.all(dp2.uca[ok] >= (dp2.dX2 * dp2.dY2).min() * (1 - 1e-12))


def test_hip_results_are_run_to_run_identical():
    """The sweep pulls in a fixed order and the pit edges are sorted before use, so two runs over the same tile must
    agree bit for bit although the tile passes, frontier appends and pit output slots are scheduled differently each
    time (size-independent property; 4096^2 bench generator, pit handling on)."""
    import warnings
    from pydem_amd import DEMProcessor
    outs = []
    for _ in range(2):
        dp = DEMProcessor.from_synthetic((4096, 4096), dict(seed=1), dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            twi = dp.calc_twi()
        outs.append((np.array(dp.mag), np.array(dp.direction), np.array(dp.section), np.array(dp.flats), np.array(dp.uca),
                     np.array(twi), np.array(dp.edge_todo), np.array(dp.edge_done)))
        del dp
    for a, b in zip(*outs):
        assert np.array_equal(a, b, equal_nan=True)
    uca = outs[0][4]
    assert np.nanmin(uca) >= 900.0 - 1e-9          # every cell carries at least its own area


@pytest.mark.parametrize('loop', ['two_cells', 'three_cells', 'two_loops'])
def test_circular_drainage_replay_vs_oracle(loop):
    """Hand-made flow fields on a level surface with cells that drain into each other: the tile passes stall, the
    reference re-seeds (dem_processing.py:951-964); the device replays that loop (K5c) -- UCA and edge flags against the
    oracle, which restates the reference's push sweep line by line."""
    import warnings
    from oracle import oracle as O
    from pydem_amd import DEMProcessor
    n, m = 9, 10
    elev = np.full((n, m), 10.0)
    direction = np.full((n, m), 1.5 * np.pi)          # everything drains south ...
    E, N, W, S = 0.0, 0.5 * np.pi, np.pi, 1.5 * np.pi
    if loop == 'two_cells':
        direction[3, 3] = E; direction[3, 4] = W
    elif loop == 'three_cells':                        # (3,3) -> (3,4) -> (4,4) ... back through a diagonal
        direction[3, 3] = E; direction[3, 4] = S; direction[4, 4] = 0.75 * np.pi      # NW: back to (3,3)
    else:
        direction[2, 2] = E; direction[2, 3] = W
        direction[5, 6] = S; direction[6, 6] = N
    mag = np.ones((n, m))
    flats = np.zeros((n, m), bool)
    o = O.OracleDEM(elev, dX=2.0, dY=3.0)
    o.mag, o.direction, o.flats = mag.copy(), direction.copy(), flats.astype(np.uint8)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        o.calc_uca()
        dp = DEMProcessor(elev=elev, dX=2.0, dY=3.0, mag=mag.copy(), direction=direction.copy(), flats=flats.copy(),
                          fill_flats=False, drain_pits_path=False)
        uca = dp.calc_uca()
    assert o.stats[0] > 1, "the case is meant to need the re-seed loop"
    _close(uca, o.uca, 'uca')
    assert np.array_equal(dp.edge_todo, o.edge_todo)
    assert np.array_equal(dp.edge_done, o.edge_done)


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_stencil_mask_path_equals_exact_path(dtype, monkeypatch):
    """The marching stencil decides the facet states by mask algebra when the band holds no NaN / huge elevation and
    facet by facet on the slopes themselves otherwise (csrc/stencil.hip); PYDEM_STENCIL_EXACT=1 forces the latter for a
    whole tile.  Both must give the same bits -- on terrain with plateaus, exact ties (integer heights) and a spacing
    that changes from row to row."""
    from pydem_amd import DEMProcessor
    rng = np.random.default_rng(11)
    n, m = 333, 517
    z = np.cumsum(np.cumsum(rng.normal(size=(n, m)), 0), 1) * 0.05
    z[40:90, 100:200] = np.round(z[40:90, 100:200])            # integer terraces: ties and flats
    z[200:230, 300:360] = z[200, 300]                            # a plateau
    z = z.astype(dtype)
    dX = np.linspace(25.0, 35.0, n - 1); dY = np.full(n - 1, 30.0)
    outs = []
    for exact in ('0', '1'):
        monkeypatch.setenv('PYDEM_STENCIL_EXACT', exact)
        dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, fill_flats=False, drain_pits_path=False)
        mag, direction = dp.calc_slopes_directions()
        outs.append((np.array(mag), np.array(direction)))
    assert np.array_equal(outs[0][0], outs[1][0], equal_nan=True), 'mag'
    assert np.array_equal(outs[0][1], outs[1][1], equal_nan=True), 'direction'
    assert (outs[0][0] == -1).any() and (outs[0][0] > 0).any()


@pytest.mark.parametrize('dtype', ['float64', 'float32'])
def test_stencil_store_split_equals_default_kernel(dtype, monkeypatch):
    """The producer / store split of the marching stencil (csrc/stencil.hip: k_stencil_march_split, PYDEM_STENCIL_SPLIT=1 -- a
    round-6 experiment that is slower and not the default, profiles/r06_stencil_split_ab.txt) computes with the same band code and
    must return the same bits as the default kernel: terraces, a plateau, a row-dependent spacing, a tile wider than eleven strips
    and a ragged last strip / chunk."""
    from pydem_amd import DEMProcessor
    rng = np.random.default_rng(12)
    n, m = 401, 1031
    z = np.cumsum(np.cumsum(rng.normal(size=(n, m)), 0), 1) * 0.05
    z[40:90, 100:200] = np.round(z[40:90, 100:200])
    z[200:230, 300:360] = z[200, 300]
    z = z.astype(dtype)
    dX = np.linspace(25.0, 35.0, n - 1); dY = np.full(n - 1, 30.0)
    outs = []
    for split in ('0', '1'):
        monkeypatch.setenv('PYDEM_STENCIL_SPLIT', split)
        dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, fill_flats=False, drain_pits_path=False)
        mag, direction = dp.calc_slopes_directions()
        outs.append((np.array(mag), np.array(direction), np.array(dp.flats)))
    for k, what in enumerate(('mag', 'direction', 'flats')):
        assert np.array_equal(outs[0][k], outs[1][k], equal_nan=True), what


def test_config1_cone256_device_vs_reference_pinned_oracle():
    """BASELINE.json config 1 on the device: the reference's 256 x 256 cone with default options (conditioning on).  The
    oracle pipeline of tests/test_oracle_golden.py::test_config1_cone256_matches_reference_checksums is bit-identical to
    the unmodified reference there; the device path must match it (masks exactly, float fields within RTOL)."""
    import warnings
    import conditioning_numpy as CN
    from oracle import oracle as O
    from pydem_amd import DEMProcessor
    nn = 256
    x, y = np.mgrid[-1:1:complex(0, nn), -1:1:complex(0, nn)]
    elev = 1 - np.sqrt(y ** 2 + x ** 2) / np.sqrt(2.)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        filled = CN.fill_flats(elev)
        drained, _, _ = CN.pit_drain_paths(np.array(filled), np.ones(nn - 1), np.ones(nn - 1))
        o = O.OracleDEM(drained, dX=1.0, dY=1.0); twi_o = o.calc_twi()
        dp = DEMProcessor(elev=elev.copy())
        twi = dp.calc_twi()
    assert np.array_equal(np.asarray(dp.elev), drained), 'conditioned surface'
    assert np.array_equal(np.asarray(dp.flats, bool), np.asarray(o.flats, bool))
    assert np.array_equal(dp.edge_todo, o.edge_todo) and np.array_equal(dp.edge_done, o.edge_done)
    _close(dp.mag, o.mag, 'mag'); _close(dp.direction, o.direction, 'direction'); _close(dp.uca, o.uca, 'uca'); _close(twi, twi_o, 'twi')


@pytest.mark.parametrize('quantised', [False, True])
def test_device_edge_set_equals_oracle_adjacency(quantised):
    """A4 directly: the device never materialises the adjacency matrix of _mk_adjacency_matrix (:1072-1153), it keeps one
    packed word per cell (in-mask, out flags, facet) plus the pit -> drain side list.  The edge set those words describe
    -- regular out-edges with weights (proportion, 1 - proportion), pit edges through the same keep-filter (:1136-1137) --
    must be the oracle's CSC triplets (themselves pinned against scipy's in tests/test_oracle_golden.py): same (source,
    target) pairs exactly, weights within RTOL, and every edge present in its target's in-mask."""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth
    n, m = 310, 270
    z = synth.fractal(n, m, seed=33, top_shift=6, n_octaves=6, zrange=(35.0 if quantised else 400.0))
    if quantised:
        z = np.rint(z)
    o = O.OracleDEM(z, dX=10.0, dY=12.0, drain_pits=True); o.calc_uca()
    indptr, indices, data = o.A
    dp = DEMProcessor(elev=z, dX=10.0, dY=12.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
    dp.calc_uca()
    words = dp._tile.graph_words().ravel()
    prop = np.asarray(dp.proportion, float).ravel()
    sec = ((words >> 12) & 7).astype(int)
    e1r = np.array([0, -1, -1, 0, 0, 1, 1, 0]); e1c = np.array([1, 0, 0, -1, -1, 0, 0, 1])
    e2r = np.array([-1, -1, -1, -1, 1, 1, 1, 1]); e2c = np.array([1, 1, -1, -1, -1, -1, 1, 1])
    cells = np.arange(n * m)
    dev = {}
    for bit, dr, dc, w in ((8, e1r, e1c, prop), (9, e2r, e2c, 1 - prop)):
        has = ((words >> bit) & 1).astype(bool)
        src = cells[has]; dst = src + dr[sec[has]] * m + dc[sec[has]]
        for a, b, x in zip(src.tolist(), dst.tolist(), w[has].tolist()):
            dev[(a, b)] = dev.get((a, b), 0.0) + x
    ps, pd, pw = dp._tile.pit_edges()
    zz = z.ravel()
    keep = ~np.isnan(pw) & (pw > 1e-8) & (zz[pd] <= zz[ps])                       # :1136-1137
    assert (((words[ps] >> 10) & 1) == 1).all() and (((words[pd[keep]] >> 11) & 1) == 1).all()
    for a, b, x in zip(ps[keep].tolist(), pd[keep].tolist(), pw[keep].tolist()):
        dev[(a, b)] = dev.get((a, b), 0.0) + x
    ref = {}
    for i in range(n * m):
        for q in range(indptr[i], indptr[i + 1]):
            ref[(i, int(indices[q]))] = float(data[q])
    assert set(dev) == set(ref), (len(dev), len(ref), sorted(set(dev) ^ set(ref))[:5])
    worst = max(abs(dev[k] - ref[k]) / max(abs(ref[k]), 1e-300) for k in ref)
    assert worst <= RTOL, worst
    # the in-mask of a target holds the bit of every regular edge into it (bit order NW N NE W E SW S SE)
    nb = {(-1, -1): 0, (-1, 0): 1, (-1, 1): 2, (0, -1): 3, (0, 1): 4, (1, -1): 5, (1, 0): 6, (1, 1): 7}
    inmask = np.zeros(n * m, np.uint32)
    for bit, dr, dc in ((8, e1r, e1c), (9, e2r, e2c)):
        has = ((words >> bit) & 1).astype(bool)
        src = cells[has]; ddr = dr[sec[has]]; ddc = dc[sec[has]]
        dst = src + ddr * m + ddc
        bits = np.array([nb[(-a, -b)] for a, b in zip(ddr.tolist(), ddc.tolist())], np.uint32)
        np.bitwise_or.at(inmask, dst, (1 << bits).astype(np.uint32))
    assert np.array_equal(inmask, words & 0xFF)
