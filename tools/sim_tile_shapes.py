#!/usr/bin/env python
"""Offline model of the tile-pass schedule for several tile SHAPES (CPU, oracle graph): per cell the pass that finishes
it and the round of its tile visit (round = 1 + longest chain of same-tile, same-pass upstream cells), then per shape:
passes, tile visits, rounds per visit, and "wave-rounds" = sum over (visit, round) of ceil(ready cells / 64) -- the
number of times a wavefront executes the round body.   sim_tile_shapes.py [size]"""
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


def schedule(tile):
    key = np.full(NN, RB + 1, np.int64)          # pass * RB + round
    deg = indeg.copy()
    frontier = np.flatnonzero(deg == 0)
    while frontier.size:
        starts, ends = indptr[frontier], indptr[frontier + 1]
        cnt = ends - starts
        e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        s = np.repeat(frontier, cnt); d = dst[e]
        same = tile[s] == tile[d]
        cand = np.where(same, key[s] + 1, (key[s] // RB + 1) * RB + 1)
        np.maximum.at(key, d, cand)
        np.subtract.at(deg, d, 1)
        frontier = np.unique(d[deg[d] == 0])
    return key // RB, key % RB


for th, tw in ((32, 32), (64, 32), (64, 64), (128, 32)):
    tile = (ii // th) * (n // tw + 1) + jj // tw
    p, r = schedule(tile)
    ntile = tile.max() + 1
    vkey = p.astype(np.int64) * ntile + tile                 # a visit
    visits = np.unique(vkey).size
    rkey = vkey * RB + r
    u, c = np.unique(rkey, return_counts=True)
    wave_rounds = int(np.ceil(c / 64).sum())
    rounds = u.size
    # rounds of the longest visit per pass, summed over the passes from 11 on (the latency-bound tail)
    vmax = {}
    for k, rr in zip(u // RB // ntile, u % RB):
        vmax[k] = max(vmax.get(k, 0), rr)
    tail = sum(v for k, v in vmax.items() if k >= 11)
    print('tile %3dx%-3d passes %3d  visits %7d (x %4d cells = %6.1f M staged)  rounds %8d (%.1f per visit)  wave-rounds %8d  '
          'cells per wave-round %.1f  longest-visit rounds summed over passes >= 11: %d' % (
              th, tw, p.max(), visits, th * tw, visits * th * tw / 1e6, rounds, rounds / visits, wave_rounds, NN / wave_rounds, tail))
