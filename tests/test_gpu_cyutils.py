"""The cyutils-compatible boundary (pydem_amd.cyfuncs.cyutils, csrc/cyutils.hip) against the oracle's
restatement of cyutils._drain_area / _drain_connections (itself pinned bit-exact through the UCA goldens),
on real flow graphs built from seeded tiles: same call sequence as the reference's _calc_uca_chunk
(dem_processing.py:879-960) and _calc_uca_chunk_update (:820-853)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _graph(shape, seed, pits):
    from oracle import oracle as O
    from pydem_amd import synth
    elev = synth.fractal(shape[0], shape[1], seed=seed, top_shift=6, n_octaves=6)
    o = O.OracleDEM(elev, dX=30.0, dY=30.0, drain_pits=pits)
    o.calc_slopes_directions(); o.build_graph()
    indptr, indices, data = o.A
    rp, ri = O.tocsr(indptr, indices, elev.size)
    return o, (indptr, indices, data, rp, ri)


@pytest.mark.parametrize('shape,seed,pits', [((120, 90), 51, False), ((200, 260), 52, True)])
def test_drain_area_matches_oracle(shape, seed, pits):
    from oracle import oracle as O
    from pydem_amd.cyfuncs import cyutils
    o, (cp, ci, cd, rp, ri) = _graph(shape, seed, pits)
    n, m = shape
    N = n * m
    insum = np.zeros(N); np.add.at(insum, ci, cd)
    ids0 = insum == 0
    rng = np.random.default_rng(seed)
    et0 = (rng.random(N) < 0.01).astype(float)

    def fresh():
        return (np.repeat(900.0, N), ids0.copy(), ids0.copy(), et0.copy(), et0.copy())
    a1, d1, i1, e1, f1 = fresh()
    O.drain_area(a1, d1, i1, cp, ci, cd, rp, ri, n, m, e1, f1)
    a2, d2, i2, e2, f2 = fresh()
    cyutils.drain_area(a2, d2, i2, cp, ci, cd, rp, ri, n, m, e2, f2)
    assert np.array_equal(d1, d2)
    # additions into a target follow the Cython order (ascending source id, no floating-point atomics): bit for bit
    assert np.array_equal(a2, a1)
    assert np.array_equal(e2, e1) and np.array_equal(f2, f1)
    a3, d3, i3, e3, f3 = fresh()
    cyutils.drain_area(a3, d3, i3, cp, ci, cd, rp, ri, n, m, e3, f3)
    assert np.array_equal(a3, a2) and np.array_equal(e3, e2)          # and run to run
    # skip_edge variant without the taint arrays (the edge-update call, :836-842)
    a1, d1, i1, _, _ = fresh(); a2, d2, i2, _, _ = fresh()
    O.drain_area(a1, d1, i1, cp, ci, cd, rp, ri, n, m, skip_edge=1)
    cyutils.drain_area(a2, d2, i2, cp, ci, cd, rp, ri, n, m, skip_edge=1)
    assert np.array_equal(d1, d2)
    assert np.array_equal(a2, a1)


@pytest.mark.parametrize('shape,seed,pits', [((150, 110), 53, True)])
def test_drain_connections_matches_oracle(shape, seed, pits):
    from oracle import oracle as O
    from pydem_amd.cyfuncs import cyutils
    o, (cp, ci, cd, rp, ri) = _graph(shape, seed, pits)
    N = shape[0] * shape[1]
    rng = np.random.default_rng(seed)
    for set_to in (False, True):
        arr = np.full(N, not set_to)
        ids = rng.random(N) < 0.002
        a1, a2 = arr.copy(), arr.copy()
        O.drain_connections(a1, ids.copy(), cp, ci, set_to=set_to)
        cyutils.drain_connections(a2, ids.copy(), cp, ci, set_to=set_to)
        assert np.array_equal(a1, a2)
        assert (a1 == set_to).sum() > 0
