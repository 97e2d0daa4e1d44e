"""edge_mode='pool': the reference's multi-worker edge schedule (pydem/process_manager.py:1214-1246) as
deterministic waves (pydem_amd/process_manager.py `_process_uca_edges_pool`), CPU tier: the per-tile
arithmetic is the oracle-backed processor, so what is tested is the schedule.

  * the reference's own acceptance test (pydem/test/test_end_to_end.py:86-149): stitched multi-tile UCA ==
    single-tile UCA on [1:-1, 1:-1] to 6 decimals, its five grid / overlap combinations, at every pool width;
  * a pit-free fractal slope (no flats, no pits: the directory flow and the single-tile flow must agree) where
    the reference's serial loop stops after one round with most edges unresolved -- the waves reach the
    single-tile answer to rounding;
  * the pm_* goldens (captured from the reference's SERIAL loop): masks and NaN patterns, and how far the
    float fields are from the serial-order result (stated per golden below).
"""
import os

import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle_processor import OracleProcessor
from test_process_manager_cpu import run_pm


def _mosaic(raster, tiles, ov, path, **kw):
    from pydem_amd import process_manager, synth
    os.makedirs(path, exist_ok=True)
    for t, (elev, bounds) in enumerate(synth.split_mosaic(raster, tiles[0], tiles[1], ov)):
        np.savez(os.path.join(path, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
    process_manager.DEBUG = True
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pm = process_manager.ProcessManager(in_path=path, elev_conditioned=True, processor_cls=OracleProcessor, **kw)
            pm.process_twi()
            return pm, pm.save_non_overlap_data()
    finally:
        process_manager.DEBUG = False


def _single(raster, **kw):
    nn = raster.shape[0]
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp = OracleProcessor(elev=raster, dX=np.ones(nn - 1), dY=np.ones(nn - 1), dX2=np.ones(nn), dY2=np.ones(nn), **kw)
        dp.calc_slopes_directions()
        dp.calc_uca()
    return dp


# the five cases of the reference's TestMultiFile (test_end_to_end.py:86-149): (ny_grid, nx_grid, overlap)
REF_CASES = [((3, 3), 2), ((4, 5), 2), ((4, 5), 3), ((3, 3), 1), ((4, 3), 1)]   # setup_case(nx_grid, ny_grid, overlap)


@pytest.mark.parametrize('tiles,ov', REF_CASES)
@pytest.mark.parametrize('n_workers', [2, 8])
def test_reference_acceptance_cases_in_pool_mode(tiles, ov, n_workers, tmp_path):
    from pydem_amd import synth
    cone = synth.cone_scaled(32)                  # NN = 32 like the reference's class (test_end_to_end.py:36)
    single = _single(cone)
    pm, compact = _mosaic(cone, tiles, ov, str(tmp_path), n_workers=n_workers)
    assert pm.edge_waves < pm.edge_rounds         # tiles did run side by side
    np.testing.assert_array_almost_equal(single.uca[1:-1, 1:-1], compact['uca'][1:-1, 1:-1], decimal=6)


@pytest.mark.parametrize('tiles,ov', [((3, 3), 2), ((2, 4), 1), ((4, 3), 3)])
def test_pool_mode_reaches_single_tile_answer_on_pit_free_slope(tiles, ov, tmp_path):
    from pydem_amd import synth
    nn = 120
    ii, jj = np.mgrid[0:nn, 0:nn]
    z = synth.fractal(nn, nn, seed=1, top_shift=7, n_octaves=7) + 40.0 * (0.7 * ii + 1.3 * jj)
    single = _single(z, drain_pits=False)
    assert not single.flats.any()
    pm, compact = _mosaic(z, tiles, ov, str(tmp_path), n_workers=8, dem_proc_kwargs={'drain_pits': False})
    a, b = compact['uca'][1:-1, 1:-1], single.uca[1:-1, 1:-1]
    np.testing.assert_allclose(a, b, rtol=1e-12, atol=0)
    assert sum(int(pm.tiles[i].edge_todo.sum()) for i in range(pm.n_inputs)) == 0
    # the serial loop of the reference gives up after its first round here (its ranking does not change)
    pm1, serial = _mosaic(z, tiles, ov, str(tmp_path / 'serial'), n_workers=1, dem_proc_kwargs={'drain_pits': False})
    assert np.nanmax(np.abs(serial['uca'][1:-1, 1:-1] - b) / b) > 0.1


@pytest.mark.parametrize('name', golden_names('pm_'))
def test_pool_mode_against_serial_order_goldens(name, tmp_path):
    """Not a parity test: the goldens are the reference's serial-order results and the schedule matters wherever
    rule :274 drops cells early (see `_process_uca_edges_pool`).  Pinned here: the run terminates, is deterministic,
    per-tile conditioning / aspect / slope are untouched by the schedule, the NaN pattern of the stitched UCA is the
    serial one, and pit-free cones agree with the serial result to rounding."""
    g = load_golden(name)
    pm, compact, order = run_pm(g, str(tmp_path / 'a'), processor_cls=OracleProcessor, n_workers=8)
    pm2, compact2, _ = run_pm(g, str(tmp_path / 'b'), processor_cls=OracleProcessor, n_workers=8)
    assert pm.edge_waves == pm2.edge_waves and pm.edge_rounds == pm2.edge_rounds
    for key in compact:
        assert np.array_equal(compact[key], compact2[key], equal_nan=True), key
    for key in ('elev', 'aspect', 'slope'):
        np.testing.assert_allclose(compact[key], g['compact_' + key], rtol=1e-12, atol=1e-13, equal_nan=True)
    assert np.array_equal(np.isnan(compact['uca']), np.isnan(g['compact_uca']))
    if name.startswith('pm_cone32') and not name.endswith('4x5_ov3'):
        np.testing.assert_allclose(compact['uca'], g['compact_uca'], rtol=1e-9, atol=0, equal_nan=True)
