"""Oracle restatement of the edge-resolution round (calc_uca(uca_init=, edge_init_data=),
reference dem_processing.py:720-771 + _calc_uca_chunk_update :778-862) against goldens captured
from the unmodified reference (g6_edge_update_*).  Bit-exact.  CPU only."""
import numpy as np
import pytest

from conftest import golden_names, load_golden
from oracle import oracle as O

KEYS = ('left', 'right', 'top', 'bottom')


def oracle_edge_round(g):
    kw = g['kwargs']
    elev = np.ascontiguousarray(g['in_elev'], np.float64)
    mag = g['in_mag'].copy()
    flats = (mag == -1).astype(np.uint8)                      # find_flats(), reference :305-306
    section, proportion = O.section_proportion(np.ascontiguousarray(g['in_direction']), flats, g['in_dX'], g['in_dY'])
    if kw.get('drain_pits', True):
        pi, pj, pp, _ = O.pit_edges(elev, flats, mag, g['in_dX'], g['in_dY'])
    else:
        pi = pj = pp = None
    A = O.adjacency(section, proportion, elev, pi, pj, pp)
    strips = lambda what: {k: g['strip_%s_%s' % (what, k)] for k in KEYS}
    return O.uca_update(elev, flats, A, strips('data'), strips('done'), strips('todo'), g['uca_init'])


@pytest.mark.parametrize('name', golden_names('g6_edge_update'))
def test_oracle_edge_update_matches_reference(name):
    g = load_golden(name)
    uca, todo, done = oracle_edge_round(g)
    assert np.array_equal(uca, g['uca'], equal_nan=True)
    assert np.array_equal(todo, g['edge_todo'])
    assert np.array_equal(done, g['edge_done'])
