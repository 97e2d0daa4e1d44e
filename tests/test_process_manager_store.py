"""The on-disk tile store of the ProcessManager drop-in (`checkpoint=True`, out_format 'npy') and the reference's
`success` resume semantics (pydem/process_manager.py:998-1007, :1027-1029, :1057-1058, :1316-1317): a directory job that
stopped after any phase continues from the stored tiles and ends with the results of an uninterrupted run.  CPU tier:
the per-tile arithmetic is the oracle-backed processor."""
import numpy as np
import pytest

from conftest import load_golden
from oracle_processor import OracleProcessor
from test_process_manager_grid import write_tiles

PHASES = ['process_elevation', 'process_aspect_slope', 'process_uca', 'process_uca_edges']


def _pm(path, out, **kw):
    from pydem_amd import process_manager
    return process_manager.ProcessManager(in_path=path, out_path=out, elev_conditioned=True, processor_cls=OracleProcessor,
                                          dem_proc_kwargs={'drain_pits': True}, **kw)


@pytest.mark.parametrize('stop_after', [0, 1, 2, 3])
@pytest.mark.parametrize('n_workers', [1, 4])
def test_resume_from_tile_store(stop_after, n_workers, tmp_path):
    from pydem_amd import process_manager
    g = load_golden('pm_fractal_2x3_ov1')
    src = str(tmp_path / 'tiles')
    write_tiles(g, src, key='elev')
    process_manager.DEBUG = True
    try:
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            ref = _pm(src, str(tmp_path / 'plain'), n_workers=n_workers)
            ref.process_twi()
            want = ref.save_non_overlap_data()
            out = str(tmp_path / 'store')
            first = _pm(src, out, n_workers=n_workers, checkpoint=True)
            first.compute_grid()
            for name in PHASES[:stop_after + 1]:
                getattr(first, name)()                         # ... and the job dies here
            calls = {'slopes': 0, 'uca': 0}
            orig_s, orig_u = OracleProcessor.calc_slopes_directions, OracleProcessor.calc_uca

            def count_s(self):
                calls['slopes'] += 1
                return orig_s(self)

            def count_u(self, uca_init=None, edge_init_data=None):
                if uca_init is None:
                    calls['uca'] += 1
                return orig_u(self, uca_init=uca_init, edge_init_data=edge_init_data)
            OracleProcessor.calc_slopes_directions, OracleProcessor.calc_uca = count_s, count_u
            try:
                again = _pm(src, out, n_workers=n_workers, checkpoint=True)
                again.process_twi()
                got = again.save_non_overlap_data()
            finally:
                OracleProcessor.calc_slopes_directions, OracleProcessor.calc_uca = orig_s, orig_u
    finally:
        process_manager.DEBUG = False
    n = first.n_inputs
    assert calls['slopes'] == (0 if stop_after >= 1 else n), "finished aspect / slope tiles were computed again"
    assert calls['uca'] == (0 if stop_after >= 2 else n), "finished first-pass UCA tiles were computed again"
    for key in want:
        assert np.allclose(got[key], want[key], rtol=1e-12, atol=1e-13, equal_nan=True), key
    tab = np.load(str(tmp_path / 'store' / 'success.npy'))
    assert tab.shape == (n, 4) and tab.all()


def test_success_table_with_worker_threads_and_several_ranks(tmp_path):
    """`_store` is called from the per-tile worker threads (tiles_in_flight > 1) and, in a multi-rank job, from several
    processes that share one `out_path`: every flag that was set must be in the published table (no lost update, no
    clash of temporary files), and a manager that starts later sees all of them."""
    from concurrent.futures import ThreadPoolExecutor
    from pydem_amd import process_manager
    g = load_golden('pm_fractal_2x3_ov1')
    src = str(tmp_path / 'tiles')
    write_tiles(g, src, key='elev')
    process_manager.DEBUG = True
    try:
        out = str(tmp_path / 'store')
        ranks = [_pm(src, out, checkpoint=True) for _ in range(2)]      # two "ranks": tile i belongs to manager i % 2
        for pm in ranks:
            pm.compute_grid()
        n = ranks[0].n_inputs
        field = np.arange(12.0).reshape(3, 4)

        def work(job):
            i, phase = job
            ranks[i % 2]._store(i, phase, {phase + '_x': field + i})
        jobs = [(i, ph) for ph in ('elev', 'aspect_slope', 'uca', 'twi') for i in range(n)]
        for _ in range(5):
            with ThreadPoolExecutor(max_workers=8) as ex:
                list(ex.map(work, jobs))
        tab = np.load(str(tmp_path / 'store' / 'success.npy'))
        assert tab.shape == (n, 4) and tab.all(), tab
        import os
        assert not [f for f in os.listdir(out) if '.tmp' in f], "temporary files left behind"
        late = _pm(src, out, checkpoint=True)
        late.compute_grid()
        assert all(late._stored(i, ph) for i, ph in jobs)
        assert np.array_equal(late._load(3, 'uca_x'), field + 3)
    finally:
        process_manager.DEBUG = False
