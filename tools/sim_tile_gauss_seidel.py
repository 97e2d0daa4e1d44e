#!/usr/bin/env python
"""Offline model (CPU, oracle graph) of tile passes that consume what OTHER tiles finished earlier in the same pass
(Gauss-Seidel between tiles) against the schedule the library runs (Jacobi: only cells stamped in earlier passes count
as final).  The tiles of a pass are visited in phases -- phase = (tile row mod K) * K + tile column mod K -- so that
a tile's neighbours of a lower phase are finished (and out of the 1024 visits in flight per XCD) when it is staged.
Per schedule: passes, tile visits, round bodies (sum over visits and rounds of ceil(ready cells / 64)).
sim_tile_gauss_seidel.py [size]"""
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
dst = indices.astype(np.int64)
indeg = np.bincount(dst, minlength=NN)
ii, jj = np.divmod(np.arange(NN), n)
RB = 1 << 12
TH = TW = 32
ntx = n // TW + 1
tile = (ii // TH) * ntx + jj // TW


def schedule(phase):
    """phase: per tile, or None for the Jacobi schedule"""
    key = np.full(NN, RB + 1, np.int64)          # pass * RB + round
    deg = indeg.copy()
    frontier = np.flatnonzero(deg == 0)
    while frontier.size:
        starts, ends = indptr[frontier], indptr[frontier + 1]
        cnt = ends - starts
        e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        s = np.repeat(frontier, cnt); d = dst[e]
        same = tile[s] == tile[d]
        nxt = (key[s] // RB + 1) * RB + 1
        if phase is None:
            cross = nxt
        else:                                     # an earlier phase of the same pass has finished: its cells are final at staging
            cross = np.where(phase[tile[s]] < phase[tile[d]], (key[s] // RB) * RB + 1, nxt)
        cand = np.where(same, key[s] + 1, cross)
        np.maximum.at(key, d, cand)
        np.subtract.at(deg, d, 1)
        frontier = np.unique(d[deg[d] == 0])
    return key // RB, key % RB


def report(name, p, r):
    ntile = tile.max() + 1
    vkey = p.astype(np.int64) * ntile + tile
    visits = np.unique(vkey).size
    u, c = np.unique(vkey * RB + r, return_counts=True)
    bodies = int(np.ceil(c / 64).sum())
    per_pass = np.bincount(np.unique(vkey) // ntile)
    print('%-34s passes %3d  visits %7d  round bodies %8d  cells per body %.1f  visits of passes 1.. : %s' % (
        name, p.max(), visits, bodies, NN / bodies, ' '.join(str(v) for v in per_pass[1:13])))


tr, tc = np.divmod(np.arange(tile.max() + 1), ntx)
report('Jacobi (as built)', *schedule(None))
for K in (2, 3, 4):
    report('Gauss-Seidel, %d x %d phases' % (K, K), *schedule((tr % K) * K + tc % K))
report('Gauss-Seidel, raster order (bound)', *schedule(np.arange(tile.max() + 1)))
# phases that keep every launch on all memory channels (the parity of a tile COLUMN is an address bit of the planes: a launch over
# the even columns only was measured at half the throughput): by tile row only, and by blocks of tiles
for K in (2, 3, 4):
    report('Gauss-Seidel, %d row phases' % K, *schedule(tr % K))
for B in (2, 4):
    report('Gauss-Seidel, 2 x 2 phases of %d x %d-tile blocks' % (B, B), *schedule(((tr // B) % 2) * 2 + (tc // B) % 2))
report('Gauss-Seidel, rows 2 x column blocks of 4', *schedule((tr % 2) * 2 + (tc // 4) % 2))
