"""Host logic of the ProcessManager drop-in (overlap-1 edge patching, strip routing, corner rules,
the reference's serial edge loop) on CPU: the per-tile arithmetic is supplied by the oracle-backed
processor in tests/oracle_processor.py, so every difference would be a ProcessManager bug.
Compared with per-tile results of the unmodified reference ProcessManager (tests/golden/pm_*.npz)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle_processor import OracleProcessor
from test_process_manager_grid import write_tiles


def run_pm(g, path, from_raw=False, **kw):
    """from_raw: start from the raw input tiles and let process_elevation condition them (reference
    calc_elev_cond :54-71); otherwise start from the reference's conditioned elevation."""
    from pydem_amd import process_manager
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}
    write_tiles(g, path, key='in_elev' if from_raw else 'elev')
    process_manager.DEBUG = True
    try:
        pm = process_manager.ProcessManager(in_path=path, dem_proc_kwargs=dkw, elev_conditioned=not from_raw, **kw)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pm.process_twi()
            compact = pm.save_non_overlap_data()
    finally:
        process_manager.DEBUG = False
    order = [int(np.argmin([np.abs(g['t%02d_bounds' % j] - pm.index[i, :4]).sum() for j in range(pm.n_inputs)]))
             for i in range(pm.n_inputs)]
    return pm, compact, order


def compare_with_golden(pm, compact, order, g, close):
    for i, j in enumerate(order):
        T = lambda key: g['t%02d_%s' % (j, key)]
        close(pm.tile_result(i, 'elev'), T('elev'), 'tile %d elev' % i)
        close(pm.tile_result(i, 'aspect'), T('aspect'), 'tile %d aspect' % i)
        close(pm.tile_result(i, 'slope'), T('slope'), 'tile %d slope' % i)
        close(pm.tile_result(i, 'uca_total'), T('uca') + T('uca_edges'), 'tile %d uca' % i)
        assert np.array_equal(pm.tile_result(i, 'edge_todo'), T('edge_todo')), 'tile %d edge_todo' % i
        assert np.array_equal(pm.tile_result(i, 'edge_done'), T('edge_done')), 'tile %d edge_done' % i
        close(pm.tile_result(i, 'twi'), T('twi'), 'tile %d twi' % i)
    for key in ('elev', 'uca', 'aspect', 'slope', 'twi'):
        close(compact[key], g['compact_' + key], 'stitched ' + key)


def _close(a, b, what):
    assert np.array_equal(np.isnan(a), np.isnan(b)), what
    assert np.allclose(a, b, rtol=1e-12, atol=1e-13, equal_nan=True), what


@pytest.mark.parametrize('name', golden_names('pm_'))
def test_directory_flow_host_logic(name, tmp_path):
    g = load_golden(name)
    pm, compact, order = run_pm(g, str(tmp_path), processor_cls=OracleProcessor)
    compare_with_golden(pm, compact, order, g, _close)


@pytest.mark.parametrize('name', ['pm_fractal_2x2_ov2', 'pm_cone32_3x3_ov1'])
def test_directory_flow_from_raw_tiles(name, tmp_path):
    """process_elevation included: raw tiles -> fill flats -> pit drain paths -> ... like the reference."""
    g = load_golden(name)
    pm, compact, order = run_pm(g, str(tmp_path), from_raw=True, processor_cls=OracleProcessor)
    compare_with_golden(pm, compact, order, g, _close)
