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
        assert np.array_equal(dp.elev, g['elev_artifacts'])
    dp = _dp(g['in_elev'].copy(), **opts)
    dp.calc_fill_flats()
    on_device = 'elev' in dp._on_device
    assert on_device or np.isnan(np.asarray(g['in_elev'], float)).any()
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
