"""GPU parity of the pit -> drain assignment (reference _mk_connectivity_pits, dem_processing.py:1269-1382)
and of everything downstream of it (patched mag/flats, UCA through non-adjacent edges, TWI).
Bar: the set of (pit, drain) pairs is bit-exact; weights follow numpy's summation order and are
compared at 1e-14 relative; float fields as in test_gpu_parity.py."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from test_gpu_parity import _close

pytestmark = pytest.mark.gpu


def _pit_cases():
    return [nm for nm in golden_names() if load_golden(nm)['kwargs'].get('drain_pits', True)]


def _check_pits(dp, pit_i, pit_j, pit_prop):
    src, dst, w = dp._tile.pit_edges()
    ref = sorted(zip(pit_i.tolist(), pit_j.tolist(), pit_prop.tolist()))
    got = sorted(zip(src.tolist(), dst.tolist(), w.tolist()))
    assert [r[:2] for r in ref] == [g[:2] for g in got], "pit -> drain assignments differ"
    if ref:
        np.testing.assert_allclose([g[2] for g in got], [r[2] for r in ref], rtol=1e-14, atol=0, equal_nan=True)


@pytest.mark.parametrize('name', _pit_cases())
def test_hip_pits_vs_reference_golden(name):
    g = load_golden(name)
    from pydem_amd import DEMProcessor
    opts = {k: v for k, v in g['kwargs'].items() if k.startswith(('drain_pits_m', 'apply_', 'uca_sat', 'twi_min'))}
    dp = DEMProcessor(elev=g['elev_final'], dX=g['in_dX'], dY=g['in_dY'], dX2=g['in_dX2'], dY2=g['in_dY2'],
                      fill_flats=False, drain_pits_path=False, drain_pits=True, **opts)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp.calc_slopes_directions()
        assert np.array_equal(dp.flats, g['flats'])
        uca = dp.calc_uca()
    _check_pits(dp, g['pit_i'], g['pit_j'], g['pit_prop'])
    assert np.array_equal(dp.section, g['section'])
    _close(dp.mag, g['mag_final'], 'mag after pit patch')
    assert np.array_equal(dp.flats, g['flats_final'])
    _close(uca, g['uca'], 'uca')
    assert np.array_equal(dp.edge_todo, g['edge_todo'])
    assert np.array_equal(dp.edge_done, g['edge_done'])
    twi = dp.calc_twi()
    _close(twi, g['twi_ret'], 'twi')


@pytest.mark.parametrize('shape,seed,quant', [((400, 333), 31, None), ((1024, 1024), 32, None), ((300, 300), 33, 25.0),
                                              ((257, 129), 34, 8.0)])
def test_hip_pits_vs_oracle_synthetic(shape, seed, quant):
    """quant: round elevations to integers over a small range -> large plateaus, many tied pits,
    regions that outgrow the 64x64 wave window (exercises the workgroup pass)."""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth
    n, m = shape
    if quant is None:
        elev = synth.fractal(n, m, seed=seed, top_shift=7, n_octaves=7)
    else:
        elev = np.rint(synth.fractal(n, m, seed=seed, top_shift=6, n_octaves=6, zrange=quant))
    kw = dict(dX=25.0 + 0.01 * np.arange(n - 1), dY=31.0 - 0.004 * np.arange(n - 1),
              dX2=25.0 + 0.01 * np.arange(n), dY2=31.0 - 0.004 * np.arange(n))
    o = O.OracleDEM(elev, drain_pits=True, **kw)
    o.calc_twi()
    dp = DEMProcessor(elev=elev, fill_flats=False, drain_pits_path=False, drain_pits=True, **kw)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        twi = dp.calc_twi()
    _check_pits(dp, o.pit_i, o.pit_j, o.pit_prop)
    assert dp.timings['n_pits_undrained'] == o.n_warn
    _close(dp.mag, o.mag, 'mag')
    assert np.array_equal(dp.flats, o.flats.astype(bool))
    assert np.array_equal(dp.section, o.section)
    _close(dp.uca, o.uca, 'uca')
    assert np.array_equal(dp.edge_todo, o.edge_todo)
    assert np.array_equal(dp.edge_done, o.edge_done)
    _close(twi, o.twi / 10, 'twi')


@pytest.mark.parametrize('seed', [1, 0])
def test_hip_pits_vs_oracle_bench_tile_4096(seed):
    """The bench generator at 4096^2 (seed 1: 400 k pit edges, ~10 % of the pits outgrow the lane pass and
    run through the wavefront pass, ~1500 pits never drain; seed 0: BASELINE config 2's exact tile, `bench.py --config 2`).
    Slope magnitude and direction cell by cell -- the 3 x 3 stencil config 2 times -- then pit assignments and UCA.
    (tools/check_pits_large.py runs the same check at the full 16384^2 size: 6 491 367 identical assignments, 5 min of
    oracle time.)"""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor
    z = O.synth_fractal(4096, 4096, seed=seed)
    o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
    o.calc_uca()
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp.calc_slopes_directions()
        dp.calc_uca()
    _close(dp.mag, o.mag, 'mag')
    _close(dp.direction, o.direction, 'direction')
    assert np.array_equal(dp.section, o.section)
    _check_pits(dp, o.pit_i, o.pit_j, o.pit_prop)
    assert dp.timings['n_pits_undrained'] == o.n_warn
    _close(dp.uca, o.uca, 'uca')


def test_hip_pits_vs_oracle_plateau_terrain_2048():
    """Config-5 style surface (int16 plateaus, conditioned on the host): borders of hundreds of cells, pits that run
    into the 300-iteration limit -- the 256x256 / 2048-cell wavefront pass and the undrained bookkeeping."""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth, conditioning
    import warnings
    n = 2048
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        z = conditioning.fill_flats(synth.srtm_int16(n, n, seed=3))
        z, _, _ = conditioning.pit_drain_paths(z, 30.0 * np.ones(n - 1), 30.0 * np.ones(n - 1))
        o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
        o.calc_uca()
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
        dp.calc_slopes_directions()
        dp.calc_uca()
    _check_pits(dp, o.pit_i, o.pit_j, o.pit_prop)
    assert dp.timings['n_pits_undrained'] == o.n_warn
    _close(dp.uca, o.uca, 'uca')


@pytest.mark.parametrize('seed', [52, 53])
def test_hip_pits_with_nodata_vs_oracle(seed):
    """NaN (nodata) cells: a lake, a block on the tile edge, isolated cells.  numpy's min propagates NaN, so a pit whose
    border touches nodata can only drain into a lower PIT cell and its region stops growing (pinned for the oracle by
    the g4_fractal_nan_* goldens of the reference); every tier of the device solver must follow."""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth
    import warnings
    n, m = 300, 420
    z = synth.fractal(n, m, seed=seed, top_shift=6, n_octaves=6)
    rng = np.random.default_rng(seed)
    z[40:90, 100:180] = np.nan
    z[0:30, 300:] = np.nan
    for _ in range(40):
        z[rng.integers(0, n), rng.integers(0, m)] = np.nan
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
        o.calc_twi()
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
        twi = dp.calc_twi()
    _check_pits(dp, o.pit_i, o.pit_j, o.pit_prop)
    assert dp.timings['n_pits_undrained'] == o.n_warn
    assert np.array_equal(dp.section, o.section) and np.array_equal(dp.flats, o.flats.astype(bool))
    _close(dp.mag, o.mag, 'mag'); _close(dp.uca, o.uca, 'uca'); _close(twi, o.twi / 10, 'twi')
    assert np.array_equal(dp.edge_todo, o.edge_todo) and np.array_equal(dp.edge_done, o.edge_done)


def _edges_and_patches(z, env):
    """pit edges (sorted by pit, drain), patched mag / flats and the tier counts of one search under `env`"""
    import os
    import warnings
    from pydem_amd import DEMProcessor
    old = {k: os.environ.get(k) for k in env}
    os.environ.update(env)
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
            dp.calc_slopes_directions()
            dp.calc_uca()
        src, dst, w = dp._tile.pit_edges()
        order = np.lexsort((dst, src))
        tm = dp.timings
        return (src[order], dst[order], w[order], dp.mag.copy(), dp.flats.copy(), dp.uca.copy(),
                {k: tm[k] for k in ('n_pits', 'n_pits_row', 'n_pits_wave', 'n_pits_big', 'n_pits_undrained', 'n_pit_edges')})
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


@pytest.mark.parametrize('kind', ['fractal', 'plateaus', 'nodata'])
def test_row_pass_equals_the_wavefront_pass(kind):
    """The row pass (16 lanes per pit, bucketed border: csrc/pits_row.inl) against the search without it, and with refill targets
    that force its corner paths: 1 (a refill nearly every round), 16 (the head fills up: entries are turned away and the
    threshold drops), 200 (everything moves to the head at once, more than 16 ties hand the pit on).  Same edges, weights
    bit for bit, same patched mag / flats, same uca; and the pass did run (PYDEM_PITS_ROW=2: also for the few pits of a small
    tile; by default it runs when the pits outnumber the resident wavefronts, as on the 16384^2 tiles of test_gpu_large_configs)."""
    from pydem_amd import synth, conditioning
    import warnings
    if kind == 'fractal':
        z = synth.fractal(1024, 1536, seed=41, top_shift=7, n_octaves=7)
    elif kind == 'plateaus':
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            z = conditioning.fill_flats(synth.srtm_int16(768, 768, seed=5))
    else:
        z = synth.fractal(700, 900, seed=43, top_shift=7, n_octaves=7)
        rng = np.random.default_rng(43)
        z[200:260, 300:420] = np.nan
        for _ in range(200):
            z[rng.integers(0, 700), rng.integers(0, 900)] = np.nan
    base = _edges_and_patches(z, {'PYDEM_PITS_ROW': '0'})
    assert base[6]['n_pits_row'] == 0 and base[6]['n_pits_wave'] > 0, base[6]
    for env in ({'PYDEM_PITS_ROW': '2'}, {'PYDEM_PITS_ROW': '2', 'PYDEM_RW_TARGET': '1'},
                {'PYDEM_PITS_ROW': '2', 'PYDEM_RW_TARGET': '16'}, {'PYDEM_PITS_ROW': '2', 'PYDEM_RW_TARGET': '200'}):
        got = _edges_and_patches(z, env)
        assert got[6]['n_pits_row'] == base[6]['n_pits_wave'] and got[6]['n_pits_wave'] < got[6]['n_pits_row'], (env, got[6])
        for k in ('n_pits', 'n_pits_undrained', 'n_pit_edges'):
            assert got[6][k] == base[6][k], (env, k, got[6], base[6])
        for a, b, what in zip(got[:6], base[:6], ('pit', 'drain', 'weight', 'mag', 'flats', 'uca')):
            assert np.array_equal(a, b, equal_nan=True), (env, what)


def test_repeated_steps_do_not_leak_device_memory():
    """A tile that is recomputed step after step (bench.py, a directory run that revisits its tiles) must reach a steady
    device footprint: every buffer of the terrain path is persistent or freed.  (Round 2 found a per-call allocation of
    the pit edge sort's temporary storage that was never freed.)"""
    import warnings
    from pydem_amd import DEMProcessor, _ffi, synth
    elev = synth.fractal(2048, 2048, seed=5, top_shift=7, n_octaves=7)
    dp = DEMProcessor(elev=elev, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
    free = []
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for _ in range(12):
            dp.calc_slopes_directions()
            dp.calc_uca()
            dp.calc_twi()
            free.append(_ffi.device_memory(0)[0])
    assert dp._tile.timings()['n_pit_edges'] > 10000
    # the first steps may still grow persistent scratch; after that a leak shows as a shrink step after step (a single late
    # one-off allocation of the runtime does not count)
    shrinks = [free[k] - free[k + 1] for k in range(3, len(free) - 1)]
    assert sum(x > 0 for x in shrinks) < 3, "device memory keeps shrinking over repeated steps: %r bytes per step" % (shrinks,)
