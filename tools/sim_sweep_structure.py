#!/usr/bin/env python
"""Offline structure of the UCA DAG on the bench terrain (CPU, oracle graph): Kahn levels, what a tile visit limited to
L local levels finishes, and what the open remainder looks like (open in-degree, chain structure).
   sim_sweep_structure.py [size]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
z = O.synth_fractal(n, n, seed=1)
o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
o.calc_slopes_directions(); o.build_graph()
indptr, indices, data = o.A            # CSC: column = from, rows = to
NN = n * n
src = np.repeat(np.arange(NN, dtype=np.int64), np.diff(indptr))
dst = indices.astype(np.int64)
indeg = np.bincount(dst, minlength=NN)
ii, jj = np.divmod(np.arange(NN), n)
print('cells', NN, 'edges', dst.size, 'pit edges (non adjacent)', int(((np.abs(ii[src] - ii[dst]) > 1) | (np.abs(jj[src] - jj[dst]) > 1)).sum()))


def out_edges(frontier):
    starts, ends = indptr[frontier], indptr[frontier + 1]
    cnt = ends - starts
    e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    return np.repeat(frontier, cnt), dst[e]


def levels(edge_ok=None):
    """Kahn level per cell (0 = source); cells that cannot finish (an in-edge is not ok) get -1."""
    lv = np.full(NN, -1, np.int32)
    deg = indeg.copy()
    frontier = np.flatnonzero(deg == 0)
    r = 0
    while frontier.size:
        lv[frontier] = r
        s, d = out_edges(frontier)
        if edge_ok is not None:
            k = edge_ok(s, d); s, d = s[k], d[k]
        np.subtract.at(deg, d, 1)
        frontier = np.unique(d[deg[d] == 0])
        r += 1
    return lv


lv = levels()
mx = lv.max()
cnt = np.bincount(lv[lv >= 0], minlength=mx + 1)
rem = NN - np.cumsum(cnt)
print('global Kahn levels', mx + 1, 'unfinished', int((lv < 0).sum()))
print('frontier sizes (first 24):', cnt[:24].tolist())
print('open after level k: ', {k: '%.3f%%' % (100.0 * rem[k] / NN) for k in (4, 9, 14, 19, 29, 49, 99, 199) if k <= mx})

for T in (32, 64, 128):
    tile = (ii // T) * (n // T + 1) + jj // T
    l1 = levels(lambda s, d: tile[s] == tile[d])     # pass 1: only in-tile edges release; a cell with an in-edge from outside never reaches zero
    fin = l1 >= 0
    c1 = np.bincount(l1[fin])
    cum = np.cumsum(c1) / NN
    print('T=%3d: pass 1 finishes %.2f%%; local depth max %d; finished within L local levels: %s' % (
        T, 100.0 * fin.mean(), l1.max() + 1, {L: '%.2f%%' % (100 * cum[min(L, len(cum)) - 1]) for L in (4, 8, 12, 16, 24, 32, 64)}))

# the open remainder after K global levels: open in-degree, chain structure
for K in (8, 12, 16, 24):
    op = lv >= K
    e_open = op[src] & op[dst]
    odeg = np.bincount(dst[e_open], minlength=NN)[op]
    outdeg = np.bincount(src[e_open], minlength=NN)[op]
    print('K=%2d: open %9d (%.3f%%)  open in-degree hist %s  open out-degree hist %s' % (
        K, op.sum(), 100.0 * op.mean(), np.bincount(odeg, minlength=5)[:6].tolist(), np.bincount(outdeg, minlength=4)[:5].tolist()))

# the tile-pass schedule (pass(c) = max over in-edges of pass(u) + [tile(u) != tile(c)]) and what is open after P passes
def tile_pass(T):
    tile = (ii // T) * (n // T + 1) + jj // T
    p = np.ones(NN, np.int32)
    deg = indeg.copy()
    frontier = np.flatnonzero(deg == 0)
    while frontier.size:
        s, d = out_edges(frontier)
        np.maximum.at(p, d, (p[s] + (tile[s] != tile[d])).astype(np.int32))
        np.subtract.at(deg, d, 1)
        frontier = np.unique(d[deg[d] == 0])
    return tile, p


for T in (32, 64):
    tile, p = tile_pass(T)
    mx = p.max()
    cells = np.bincount(p, minlength=mx + 1)[1:]
    print('T=%d passes %d; cells per pass (first 16) %s' % (T, mx, cells[:16].tolist()))
    for P in (2, 4, 6, 8, 10, 14, 20):
        op = p > P
        if not op.any(): break
        tl = np.unique(tile[op])
        e_open = op[src] & op[dst]
        cross = e_open & (tile[src] != tile[dst])
        # inlets of a tile: distinct open source cells outside the tile with an edge into it
        key = tile[dst[cross]] * NN + src[cross]
        inl = np.unique(key) // NN
        n_in = np.bincount(np.searchsorted(tl, inl), minlength=tl.size)
        n_open = np.bincount(np.searchsorted(tl, tile[op]), minlength=tl.size)
        odeg = np.bincount(dst[e_open], minlength=NN)[op]
        print('  after pass %2d: open %8d (%.3f%%) in %6d tiles (%.1f%% of tiles); open cells/tile mean %.1f p95 %d max %d; inlets/tile mean %.2f p50 %d p90 %d p99 %d max %d; open in-degree hist %s' % (
            P, op.sum(), 100.0 * op.mean(), tl.size, 100.0 * tl.size / ((n // T) ** 2), n_open.mean(), np.percentile(n_open, 95), n_open.max(),
            n_in.mean(), np.percentile(n_in, 50), np.percentile(n_in, 90), np.percentile(n_in, 99), n_in.max(), np.bincount(odeg, minlength=5)[:5].tolist()))
