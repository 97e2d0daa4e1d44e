"""Host logic of the ProcessManager drop-in (overlap-1 edge patching, strip routing, corner rules,
the reference's serial edge loop) on CPU: the per-tile arithmetic is supplied by the oracle-backed
processor in tests/oracle_processor.py, so every difference would be a ProcessManager bug.
Compared with per-tile results of the unmodified reference ProcessManager (tests/golden/pm_*.npz)."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle_processor import OracleProcessor
from test_process_manager_grid import write_tiles


def run_pm(g, path, **kw):
    from pydem_amd import process_manager
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}
    write_tiles(g, path, key='elev')
    process_manager.DEBUG = True
    try:
        pm = process_manager.ProcessManager(in_path=path, dem_proc_kwargs=dkw, elev_conditioned=True, **kw)
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
