"""A short slice of the randomised soaks (tools/soak_parity.py, tools/soak_pm.py): random tiles and random mosaics,
device path against the CPU oracle.  The full soaks run for minutes on the GPU box; these
fixed case ranges keep the generators and the comparison alive in the regular GPU tier."""
import os
import sys

import pytest

pytestmark = pytest.mark.gpu

TOOLS = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools')
if TOOLS not in sys.path:
    sys.path.insert(0, TOOLS)


@pytest.mark.parametrize('first', [0, 300000])
def test_random_tiles_match_the_oracle(first):
    import soak_parity
    for k in range(first, first + 25):
        rec, errs = soak_parity.run_case(k)
        assert not errs, (rec, errs)


def test_interior_winner_exactly_on_the_table_angle():
    """Soak case 900039 (int32 DEM, dX = 1, dY = 17): a cell whose winning facet is interior with slopes in the exact ratio of
    the spacings (s2 * d1 == s1 * d2: not clamped, pydem/dem_processing.py:1973) has the table angle as its direction, bit for
    bit -- the device's own arctangent was an ulp off there, which moved the cell into the neighbouring section
    (csrc/stencil.hip: the tie test on the winner)."""
    import soak_parity
    rec, errs = soak_parity.run_case(900039)
    assert not errs, (rec, errs)
    # the same situation by construction: dY / dX = 3, planes whose facet slopes are in exactly that ratio in many cells
    import warnings
    import numpy as np
    from oracle import oracle as O
    from pydem_amd import DEMProcessor
    rng = np.random.default_rng(7)
    ii, jj = np.mgrid[0:160, 0:200]
    for sx, sy in ((4, 36), (2, 18), (12, 4), (6, 2)):
        z = (10000 - sx * jj - sy * ii + rng.integers(0, 2, ii.shape) * 0).astype(np.int32)
        z[40:60, 50:90] += rng.integers(-3, 4, (20, 40)).astype(np.int32)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            o = O.OracleDEM(z, dX=1.0, dY=3.0, drain_pits=False); o.calc_slopes_directions(); o.calc_uca()
            dp = DEMProcessor(elev=z, dX=1.0, dY=3.0, fill_flats=False, drain_pits_path=False, drain_pits=False)
            dp.calc_slopes_directions(); dp.calc_uca()
        assert np.array_equal(dp.section, o.section), (sx, sy, int((np.asarray(dp.section) != o.section).sum()))
        assert np.allclose(dp.direction, o.direction, rtol=1e-12, atol=1e-15, equal_nan=True)


def test_random_mosaics_match_the_oracle_backed_directory_flow():
    import numpy as np
    import soak_pm
    from oracle_processor import OracleProcessor
    for k in range(30):
        rec, z, ny, nx, ov, dkw = soak_pm.make_case(k)
        ref = soak_pm.run(z, ny, nx, ov, dkw, OracleProcessor)
        dev = soak_pm.run(z, ny, nx, ov, dkw, None)
        assert dev.edge_rounds == ref.edge_rounds, rec
        for i in range(ref.n_inputs):
            zero_area = np.abs(np.asarray(ref.tile_result(i, 'uca_total'), float)) < 1e-9
            for key in ('aspect', 'slope', 'uca_total', 'twi'):
                a, b = np.asarray(dev.tile_result(i, key), float), np.asarray(ref.tile_result(i, key), float)
                if key == 'twi':
                    a = np.where(zero_area, 0.0, a); b = np.where(zero_area, 0.0, b)
                assert np.array_equal(np.isnan(a), np.isnan(b)), (rec, i, key)
                assert np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True), (rec, i, key)
            for key in ('edge_todo', 'edge_done'):
                assert np.array_equal(dev.tile_result(i, key), ref.tile_result(i, key)), (rec, i, key)


def test_circular_drainage_is_replayed_like_the_reference():
    """Mosaic 122733 of the soak: the overlap-1 patch closes a two-cell loop in one tile; the reference / oracle re-seed
    the stalled sweep (dem_processing.py:951-964, cyutils.pyx:119-187), the device replays that loop over the unfinished
    cells.  Device flow against the oracle-backed flow, like every other mosaic of the soak."""
    import numpy as np
    import soak_pm
    from oracle_processor import OracleProcessor
    rec, z, ny, nx, ov, dkw = soak_pm.make_case(122733)
    ref = soak_pm.run(z, ny, nx, ov, dkw, OracleProcessor)
    dev = soak_pm.run(z, ny, nx, ov, dkw, None)
    assert dev.edge_rounds == ref.edge_rounds
    for i in range(ref.n_inputs):
        for key in ('uca_total', 'twi'):
            a, b = np.asarray(dev.tile_result(i, key), float), np.asarray(ref.tile_result(i, key), float)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (i, key)
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True), (i, key)
        for key in ('edge_todo', 'edge_done'):
            assert np.array_equal(dev.tile_result(i, key), ref.tile_result(i, key)), (i, key)


def _same_flow(dev, ref, rec):
    import numpy as np
    assert (dev.edge_rounds, dev.edge_waves) == (ref.edge_rounds, ref.edge_waves), rec
    for i in range(ref.n_inputs):
        zero_area = np.abs(np.asarray(ref.tile_result(i, 'uca_total'), float)) < 1e-9
        for key in ('uca_total', 'twi'):
            a, b = np.asarray(dev.tile_result(i, key), float), np.asarray(ref.tile_result(i, key), float)
            if key == 'twi':
                a = np.where(zero_area, 0.0, a); b = np.where(zero_area, 0.0, b)
            assert np.array_equal(np.isnan(a), np.isnan(b)), (rec, i, key)
            assert np.allclose(a, b, rtol=1e-9, atol=1e-12, equal_nan=True), (rec, i, key)
        for key in ('edge_todo', 'edge_done'):
            assert np.array_equal(dev.tile_result(i, key), ref.tile_result(i, key)), (rec, i, key)


@pytest.mark.parametrize('compact', ['1', '0'])
def test_random_mosaics_in_pool_mode(compact, monkeypatch):
    """The pool schedule on the device (waves, incremental rounds, edge board) against the same schedule with the numpy
    strip rules and the oracle processor, on the first mosaics of the soak (SOAK_POOL=1) and on three it found.  94: a NaN seed
    sits above an edge cell that adopts its neighbour's finished value a wave later -- the reference's rounds keep the NaN
    (it is absorbing in `area_edges - uca`), so the incremental rounds flood it at once (k_cinc_nan_flood /
    k_einc_nan_flood; compact = '0' forces the cell-indexed form); 53: ... but not into the other seeds of the same round;
    264: a seed that still waits for its own upstream cells is re-initialised when its neighbour's copy has moved on; 8483: a
    tile with a drainage loop (the counts never reach zero there) takes the plain rounds."""
    import numpy as np
    import soak_pm
    from oracle_processor import OracleProcessor
    if compact == '0':
        monkeypatch.setenv('PYDEM_EINC_COMPACT_MAX', '0')
    monkeypatch.setattr(soak_pm, 'POOL', True)
    for k in list(range(12)) + [53, 94, 264, 8483]:
        rec, z, ny, nx, ov, dkw = soak_pm.make_case(k)
        width = int(np.random.default_rng(77 + k + 1).choice([2, 3, 8]))
        ref = soak_pm.run(z, ny, nx, ov, dkw, OracleProcessor, width)
        dev = soak_pm.run(z, ny, nx, ov, dkw, None, width)
        _same_flow(dev, ref, rec)


def test_device_memory_reaches_a_steady_state():
    """tools/leak_probe.py: tiles (plain, conditioned) and directory runs (serial order, pool mode) created and dropped over
    and over must not keep shrinking the free device memory."""
    import subprocess
    r = subprocess.run([sys.executable, os.path.join(TOOLS, 'leak_probe.py'), '6'], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'STEADY' in r.stdout
