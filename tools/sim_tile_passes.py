#!/usr/bin/env python
"""Offline model of the tile-pass schedule (CPU, oracle graph): for every cell the pass in which a schedule with
square tiles of edge T would finish it (pass(c) = max over in-edges u->c of pass(u) + [tile(u) != tile(c)], sources 1),
and per pass the cells finished and the 32x32 tiles that have work.  Compares T = 32 with T = 64 (2x2 groups of tiles
that hand cells over on chip) and with 64 in the first two passes only.   sim_tile_passes.py [size]"""
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
order_ptr = indptr
ii, jj = np.divmod(np.arange(NN), n)


def passes(tile_of_fn):
    tile = tile_of_fn()
    p = np.ones(NN, np.int32)
    deg = indeg.copy()
    frontier = np.flatnonzero(deg == 0)
    while frontier.size:
        # out-edges of the frontier
        starts, ends = indptr[frontier], indptr[frontier + 1]
        cnt = ends - starts
        e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        s = np.repeat(frontier, cnt); d = dst[e]
        cand = p[s] + (tile[s] != tile[d])
        np.maximum.at(p, d, cand.astype(np.int32))
        np.subtract.at(deg, d, 1)
        frontier = np.unique(d[deg[d] == 0])
    return p


t32 = (ii // 32) * (n // 32 + 1) + jj // 32
p32 = passes(lambda: t32)
p64 = passes(lambda: (ii // 64) * (n // 64 + 1) + jj // 64)
for name, p in (('T=32', p32), ('T=64 groups', p64)):
    mx = p.max()
    cells = np.bincount(p, minlength=mx + 1)[1:]
    visits = [np.unique(t32[p == k]).size for k in range(1, mx + 1)]
    print('%-12s passes %3d  tile visits %8d   cells/pass (first 8): %s   visits/pass (first 10): %s' % (
        name, mx, sum(visits), cells[:8].tolist(), visits[:10]))
