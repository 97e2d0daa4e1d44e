"""GPU parity of one edge-resolution round (reference calc_uca(uca_init=, edge_init_data=),
dem_processing.py:720-771 / _calc_uca_chunk_update :778-862) against the reference goldens
(g6_edge_update_*) and against the oracle on larger seeded tiles."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from test_gpu_parity import _close

pytestmark = pytest.mark.gpu
KEYS = ('left', 'right', 'top', 'bottom')


def _strips(g, what):
    return {k: g['strip_%s_%s' % (what, k)] for k in KEYS}


@pytest.mark.parametrize('name', golden_names('g6_edge_update'))
def test_hip_edge_update_vs_reference_golden(name):
    g = load_golden(name)
    kw = g['kwargs']
    from pydem_amd import DEMProcessor
    # exactly how process_manager.calc_uca_ec builds the processor (:227-240): stored elev/aspect/slope
    dp = DEMProcessor(elev=g['in_elev'], dX=g['in_dX'], dY=g['in_dY'], dX2=g['in_dX2'], dY2=g['in_dY2'],
                      mag=g['in_mag'], direction=g['in_direction'], fill_flats=False, drain_pits_path=False,
                      drain_pits=kw.get('drain_pits', True))
    dp.find_flats()
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        uca = dp.calc_uca(uca_init=g['uca_init'], edge_init_data=[_strips(g, 'data'), _strips(g, 'done'), _strips(g, 'todo')])
    _close(uca, g['uca'], 'uca after edge round')
    assert np.array_equal(dp.edge_todo, g['edge_todo'])
    assert np.array_equal(dp.edge_done, g['edge_done'])


@pytest.mark.parametrize('shape,seed,pits', [((300, 420), 41, False), ((512, 512), 42, True)])
def test_hip_edge_update_vs_oracle(shape, seed, pits):
    """Same tile handle: pydem_uca first (graph stays resident), then an edge round on top of it."""
    from oracle import oracle as O
    from test_oracle_edge_update import KEYS as K2  # noqa: F401
    from pydem_amd import DEMProcessor, synth
    n, m = shape
    elev = synth.fractal(n, m, seed=seed, top_shift=7, n_octaves=7)
    o = O.OracleDEM(elev, dX=30.0, dY=30.0, drain_pits=pits)
    o.calc_uca()
    rng = np.random.default_rng(seed)
    sides = {'left': (slice(None), 0), 'right': (slice(None), -1), 'top': (0, slice(None)), 'bottom': (-1, slice(None))}
    data, dn, td = {}, {}, {}
    for k, sl in sides.items():
        L = o.uca[sl].size
        data[k] = np.nan_to_num(o.uca[sl], nan=900.0) + rng.random(L) * 1e6 * (rng.random(L) < 0.5)
        dn[k] = rng.random(L) < 0.7
        td[k] = o.edge_todo[sl].copy()
    uca_ref, todo_ref, done_ref = O.uca_update(o.elev, o.flats, o.A, data, dn, td, o.uca)
    dp = DEMProcessor(elev=elev, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=pits)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        uca0 = dp.calc_uca()
        _close(uca0, o.uca, 'uca (first pass)')
        uca = dp.calc_uca(uca_init=uca0, edge_init_data=[data, dn, td])
    _close(uca, uca_ref, 'uca after edge round')
    assert np.array_equal(dp.edge_todo, todo_ref)
    assert np.array_equal(dp.edge_done, done_ref)


@pytest.mark.parametrize('shape,seed,pits', [((300, 420), 41, False), ((512, 512), 42, True), ((257, 190), 7, True)])
def test_incremental_edge_rounds_vs_oracle_rounds(shape, seed, pits):
    """pydem_uca_edge_round_inc (the fix-up state persists between rounds, every cell is processed once) against the
    oracle's plain rounds over a series of four rounds in which more and more of the neighbour strips are finished:
    masks after every round exactly, areas of finished cells after every round, all areas after the final flush.  Both
    sides use the pool schedule's seed rule (a not-done edge cell adopts a finished neighbour value,
    process_manager._adopt_finished)."""
    from oracle import oracle as O
    from pydem_amd import DEMProcessor, synth
    n, m = shape
    elev = synth.fractal(n, m, seed=seed, top_shift=7, n_octaves=7)
    o = O.OracleDEM(elev, dX=30.0, dY=30.0, drain_pits=pits)
    o.calc_uca()
    rng = np.random.default_rng(seed)
    sides = {'left': (slice(None), 0), 'right': (slice(None), -1), 'top': (0, slice(None)), 'bottom': (-1, slice(None))}
    value = {k: np.nan_to_num(o.uca[sl], nan=900.0) + rng.random(o.uca[sl].size) * 1e5 for k, sl in sides.items()}
    order = {k: rng.random(o.uca[sl].size) for k, sl in sides.items()}
    dp = DEMProcessor(elev=elev, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=pits)
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp.run_uca()
        uca_o, todo_o, done_o = o.uca.copy(), o.edge_todo.astype(bool), o.edge_done.astype(bool)
        assert np.array_equal(dp.edge_done, done_o)
        for rnd, frac in enumerate([0.25, 0.5, 0.5, 0.8, 1.0]):
            dn = {k: order[k] < frac for k in sides}
            td = {k: todo_o[sl].copy() for k, sl in sides.items()}
            if rnd == 2:
                for k in td:                                    # some 'todo' flags dropped without a value (rule :274)
                    td[k] &= rng.random(td[k].size) < 0.7
            td_o = {k: td[k] | (dn[k] & ~done_o[sl]) for k, sl in sides.items()}
            uca_o, todo_o, done_o = O.uca_update(o.elev, o.flats, o.A, value, dn, td_o, uca_o)
            todo_o = todo_o.astype(bool); done_o = done_o.astype(bool)
            dp.run_uca(edge_init_data=[value, dn, td], uca_resident=True, incremental=True)
            assert np.array_equal(dp.edge_todo, todo_o), rnd
            assert np.array_equal(dp.edge_done, done_o), rnd
            lines = {k: dp.get_line('uca', 1 if k in ('left', 'right') else 0, 0 if k in ('left', 'top') else -1) for k in sides}
            for k, sl in sides.items():
                ok = done_o[sl]
                assert np.allclose(lines[k][ok], uca_o[sl][ok], rtol=1e-9, atol=1e-12, equal_nan=True), (rnd, k)
        dp.flush_edge_rounds()
    _close(dp.uca, uca_o, 'uca after the flush')
