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
