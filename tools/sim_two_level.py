#!/usr/bin/env python
"""Offline model of a TWO-LEVEL (tile -> perimeter) solve for the tail of the UCA sweep (CPU, oracle graph).

Today: tile passes.  A pass visits a 32 x 32 tile and finishes what its neighbours released in EARLIER passes, so a flow
path advances one tile per pass (82 passes at 16384^2, `profiles/r04_sweep_passes_dense0.txt`).

Modelled here: run the tile passes up to pass P (P = 1, 2, 3), then for every tile that still has open cells
  (a) ONE symbolic visit: every open cell as  constant + sum_j coef_j * x_j  over the tile's open INLETS x_j (open cells of
      other tiles with an edge into the tile, pit -> drain edges included), emitting one record per OUTLET (open cell with
      an edge that leaves the tile): constant + its non-zero coefficients,
  (b) a Kahn solve on the coarse graph of the outlets only (edge inlet -> outlet when the coefficient is non-zero),
  (c) ONE numeric visit per tile with every inlet final (back substitution): all open cells finish.

Printed per P: open cells / tiles, inlets and outlets per tile, the LOCAL depth of a full visit (its rounds), entries of the
symbolic vectors (dense n_open x J and sparse), coarse nodes / edges / depth, and the work in the units the device pays in:
tile visits and executed round bodies (wave-rounds), against the same units for the tile passes > P that (a)-(c) replace.
The cost line uses two throughput constants fitted on the 16384^2 trace (r02_tile_pass_phases / r04_sweep_passes_dense0:
passes 1-2 = 0.524 M visits, 11.45 M rounds, 12.5 ms; passes >= 3 = 1.26 M visits, 14.0 M rounds, 19.2 ms
=> 6.3 us per visit, 0.80 us per round, whole-GPU throughput).

    sim_two_level.py [size] [tile]
"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
TS = int(sys.argv[2]) if len(sys.argv) > 2 else 32
z = O.synth_fractal(n, n, seed=1)
o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
o.calc_slopes_directions(); o.build_graph()
indptr, indices, data = o.A            # CSC: column = from, rows = to
NN = n * n
dst = indices.astype(np.int64)
src = np.repeat(np.arange(NN, dtype=np.int64), np.diff(indptr))
ii, jj = np.divmod(np.arange(NN), n)
tiles_x = (n + TS - 1) // TS
tile = (ii // TS) * tiles_x + jj // TS
ntile = int(tile.max()) + 1
RB = 1 << 12
A_VISIT, A_ROUND = 6.3e-6, 0.80e-6     # ms per visit / per round at whole-GPU throughput (see the docstring)
SCALE = (16384 / n) ** 2


def longest_path(es, ed, restart):
    """Kahn over the edge list (es -> ed): key = 1 for sources; key[d] = max(key[s] + 1) over plain edges and
    (key[s] // RB + 1) * RB + 1 over `restart` edges (a tile crossing starts a new pass)."""
    key = np.full(NN, RB + 1, np.int64)
    deg = np.bincount(ed, minlength=NN)
    order = np.argsort(es, kind='stable')
    es, ed, restart = es[order], ed[order], restart[order]
    ptr = np.searchsorted(es, np.arange(NN + 1))
    touched = np.zeros(NN, bool); touched[es] = True; touched[ed] = True
    frontier = np.flatnonzero((deg == 0) & touched)
    while frontier.size:
        starts, ends = ptr[frontier], ptr[frontier + 1]
        cnt = ends - starts
        e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
        s = np.repeat(frontier, cnt); d = ed[e]
        cand = np.where(restart[e], (key[s] // RB + 1) * RB + 1, key[s] + 1)
        np.maximum.at(key, d, cand)
        np.subtract.at(deg, d, 1)
        frontier = np.unique(d[deg[d] == 0])
    return key


def wave_rounds(visit_key, rnd, sel):
    u, c = np.unique(visit_key[sel] * RB + rnd[sel], return_counts=True)
    return int(np.unique(visit_key[sel]).size), int(u.size), int(np.ceil(c / 64).sum())


cross = tile[src] != tile[dst]
key = longest_path(src, dst, cross)
p, r = key // RB, key % RB
print('size %d  tile %d  cells %d  edges %d  passes %d' % (n, TS, NN, src.size, p.max()))
vkey = p * ntile + tile
for P in (1, 2, 3):
    opn = p > P
    v_old, r_old, w_old = wave_rounds(vkey, r, opn)
    n_open = int(opn.sum())
    t_open = np.unique(tile[opn])
    eo = opn[src]                                    # an open source makes its target open
    inl = eo & cross
    # inlets per tile = distinct (tile of target, source); outlets = distinct sources of crossing open edges
    pair = np.unique(tile[dst[inl]] * NN + src[inl])
    J = np.bincount(pair // NN, minlength=ntile)
    outlets = np.unique(src[inl])
    n_out = np.bincount(tile[outlets], minlength=ntile)
    # local depth: longest chain of open cells inside the tile
    loc = eo & ~cross
    lkey = longest_path(src[loc], dst[loc], np.zeros(int(loc.sum()), bool))
    ld = np.where(opn, lkey % RB, 0)
    depth_t = np.zeros(ntile, np.int64); np.maximum.at(depth_t, tile[opn], ld[opn])
    v_new, r_new, w_new = wave_rounds(tile, ld, opn)
    # symbolic masks: which inlets reach a cell through open cells of its own tile (Python ints as bit sets)
    mask = {}
    inlet_idx = {}
    cnt_t = {}
    for q in pair:                                   # pair is sorted by tile: running index per tile
        t, s = divmod(int(q), NN)
        k = cnt_t.get(t, 0); cnt_t[t] = k + 1
        inlet_idx[(t, s)] = k
    for s_, d_ in zip(src[inl].tolist(), dst[inl].tolist()):
        mask[d_] = mask.get(d_, 0) | (1 << inlet_idx[(int(tile[d_]), s_)])
    ls, ldst = src[loc], dst[loc]
    order = np.argsort(ld[ldst], kind='stable')      # by local depth of the target: sources are complete when read
    for s_, d_ in zip(ls[order].tolist(), ldst[order].tolist()):
        ms = mask.get(s_, 0)
        if ms:
            mask[d_] = mask.get(d_, 0) | ms
    sparse_entries = sum(bin(v).count('1') for v in mask.values())
    ent = np.zeros(NN, np.int64)
    for c_, v in mask.items():
        ent[c_] = bin(v).count('1')
    pool_t = np.bincount(tile[opn], weights=ent[opn], minlength=ntile).astype(np.int64)
    open_in = np.bincount(dst[eo], minlength=NN)      # open in-edges per open cell (in-tile + inlets)
    resolved = opn & (ent == 0)                       # no inlet upstream inside the tile: finished numerically by the symbolic visit itself
    dense_entries = int((np.bincount(tile[opn], minlength=ntile) * J).sum())
    nnz = [bin(mask.get(int(c), 0)).count('1') for c in outlets]
    # coarse Kahn depth: level(outlet) = 1 + max level(inlets it depends on); global topological order = (pass, round)
    inl_of_tile = {}
    for q in pair:
        t, s = divmod(int(q), NN)
        inl_of_tile.setdefault(t, []).append(s)
    lvl = {}
    for c in outlets[np.argsort(key[outlets], kind='stable')].tolist():
        mk = mask.get(c, 0); t = int(tile[c]); best = 0
        lst = inl_of_tile.get(t, [])
        k = 0
        while mk:
            if mk & 1:
                best = max(best, lvl[lst[k]])
            mk >>= 1; k += 1
        lvl[c] = best + 1
    lv = np.array(list(lvl.values()))
    cdepth = int(lv.max()) if lv.size else 0
    lvl_hist = np.bincount(lv)
    big = int((lvl_hist > 4096).sum())
    nt = t_open.size
    Jt, Ot, Dt, Nt = J[t_open], n_out[t_open], depth_t[t_open], np.bincount(tile[opn], minlength=ntile)[t_open]
    q = lambda a: '%.1f / %d / %d / %d' % (a.mean(), np.percentile(a, 50), np.percentile(a, 95), a.max())
    print('\n== tile passes 1..%d, then two-level' % P)
    print('open cells %d (%.1f %%) in %d of %d tiles; per open tile (mean / p50 / p95 / max): cells %s  inlets %s  outlets %s  local depth %s'
          % (n_open, 100.0 * n_open / NN, nt, ntile, q(Nt), q(Jt), q(Ot), q(Dt)))
    print('tiles with > 256 open cells: %d   with > 16 inlets: %d   with > 32 inlets: %d   with > 64: %d'
          % ((Nt > 256).sum(), (Jt > 16).sum(), (Jt > 32).sum(), (Jt > 64).sum()))
    print('symbolic entries: dense (open cells x inlets of the tile) %d = %.1f per open cell; sparse (inlets that reach the cell) %d = %.2f per open cell'
          % (dense_entries, dense_entries / max(n_open, 1), sparse_entries, sparse_entries / max(n_open, 1)))
    eo_ = ent[opn]
    print('entries per open cell: 0: %.1f %%  1: %.1f %%  2: %.1f %%  3-4: %.1f %%  5-8: %.1f %%  > 8: %.1f %%  max %d;  open in-edges per open cell: 1: %.1f %%  2: %.1f %%  >= 3: %.1f %%'
          % (100 * (eo_ == 0).mean(), 100 * (eo_ == 1).mean(), 100 * (eo_ == 2).mean(), 100 * ((eo_ >= 3) & (eo_ <= 4)).mean(),
             100 * ((eo_ >= 5) & (eo_ <= 8)).mean(), 100 * (eo_ > 8).mean(), eo_.max(),
             100 * (open_in[opn] == 1).mean(), 100 * (open_in[opn] == 2).mean(), 100 * (open_in[opn] >= 3).mean()))
    pt = pool_t[t_open]
    print('entries per tile (pool): %s;  tiles over a pool of 512 / 768 / 1024 / 2048 entries: %d / %d / %d / %d;  tiles with > 512 open cells: %d;  > 64 outlets: %d'
          % (q(pt), (pt > 512).sum(), (pt > 768).sum(), (pt > 1024).sum(), (pt > 2048).sum(), (Nt > 512).sum(), (Ot > 64).sum()))
    for rc, pc in ((256, 768), (256, 1024), (512, 2048)):
        fb = (Nt > rc) | (pt > pc) | (Jt > 64) | (Ot > 64)
        print('   caps %d cells / %d entries / 64 inlets / 64 outlets: %d tiles (%.1f %%) stay numeric, holding %.1f %% of the open cells'
              % (rc, pc, fb.sum(), 100.0 * fb.mean(), 100.0 * Nt[fb].sum() / max(Nt.sum(), 1)))
    print('cells the symbolic visit finishes numerically (no open inlet upstream in the tile): %d = %.1f %% of the open cells' % (resolved.sum(), 100.0 * resolved.sum() / max(n_open, 1)))
    print('coarse graph: nodes (outlets) %d = %.2f %% of the cells, edges %d (%.2f per node), Kahn depth %d (levels with > 4096 nodes: %d); tile passes it replaces: %d'
          % (outlets.size, 100.0 * outlets.size / NN, sum(nnz), sum(nnz) / max(outlets.size, 1), cdepth, big, p.max() - P))
    print('work  tile passes > %d : visits %8d  rounds %9d  wave-rounds %9d  (%.1f rounds per visit)' % (P, v_old, r_old, w_old, r_old / max(v_old, 1)))
    print('work  ONE full visit   : visits %8d  rounds %9d  wave-rounds %9d  (%.1f rounds per visit)' % (v_new, r_new, w_new, r_new / max(v_new, 1)))
    t_old = SCALE * (A_VISIT * v_old + A_ROUND * r_old)
    t_new1 = SCALE * (A_VISIT * v_new + A_ROUND * r_new)
    print('cost at 16384^2 with the fitted constants (generic rounds): tile passes > %d  %.1f ms;  one full visit %.1f ms, two (symbolic + numeric) %.1f ms'
          % (P, t_old, t_new1, 2 * t_new1))
    for f in (0.5, 0.3):
        print('   ... with LDS-resident rounds at %.1f x the generic round cost: two visits %.1f ms' % (f, 2 * SCALE * (A_VISIT * v_new + f * A_ROUND * r_new)))
