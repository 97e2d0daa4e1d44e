#!/usr/bin/env python
"""Offline model (CPU, oracle graph) of two prologues of the UCA sweep and of what they leave to the tile passes:
  * dense Kahn levels: K full-grid level kernels (a cell finishes in level k when all its sources finished before k);
  * directional marching sweeps: a wavefront marches the rows (or columns) of a strip of LANES cells, a cell of the row
    in hand finishes when its sources are final -- those of the row behind may have finished one step earlier in the
    SAME sweep (registers / lane shifts), those of the same row in an earlier iteration of the row step (ITERS), those
    ahead only in an earlier sweep; pit in-edges only count when the source finished in an earlier sweep.
For each prologue: open cells left, then the tile-pass schedule (32x32 tiles) on the remainder: passes, visits, rounds,
wave-rounds (sum over visits and rounds of ceil(ready / 64)).    sim_dir_sweeps.py [size] [lanes] [seg] [iters]"""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
LANES = int(sys.argv[2]) if len(sys.argv) > 2 else 62
SEG = int(sys.argv[3]) if len(sys.argv) > 3 else 256
ITERS = int(sys.argv[4]) if len(sys.argv) > 4 else 2
z = O.synth_fractal(n, n, seed=1)
o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
o.calc_slopes_directions(); o.build_graph()
indptr, indices, data = o.A            # CSC: column = from, rows = to
NN = n * n
src = np.repeat(np.arange(NN, dtype=np.int64), np.diff(indptr))
dst = indices.astype(np.int64)
ii, jj = np.divmod(np.arange(NN), n)
is_pit_src = np.zeros(NN, bool)
if o.pit_i is not None and o.pit_i.size:
    is_pit_src[o.pit_i] = True
pit_e = is_pit_src[src]
print('cells', NN, 'edges', dst.size, 'pit edges', int(pit_e.sum()))
DI = np.array([-1, -1, -1, 0, 0, 1, 1, 1]); DJ = np.array([-1, 0, 1, -1, 1, -1, 0, 1])
# regular edges -> in-mask bits of the destination
rs, rd = src[~pit_e], dst[~pit_e]
di, dj = ii[rs] - ii[rd], jj[rs] - jj[rd]
assert (np.abs(di) <= 1).all() and (np.abs(dj) <= 1).all()
code = (di + 1) * 3 + (dj + 1)
bit = np.where(code > 4, code - 1, code)
inmask = np.zeros(NN, np.uint8)
np.bitwise_or.at(inmask, rd, (1 << bit).astype(np.uint8))
inmask = inmask.reshape(n, n)
ps, pd = src[pit_e], dst[pit_e]
indeg = np.bincount(dst, minlength=NN)
RB = 1 << 12


def tile_schedule(fin):
    """tile passes (32 x 32) on the cells that are not final: per cell pass and round like tools/sim_tile_shapes.py"""
    tile = (ii // 32) * (n // 32 + 1) + jj // 32
    key = np.full(NN, RB + 1, np.int64)
    deg = indeg.copy()
    finl = np.flatnonzero(fin)
    # remove the final cells: their out-edges are satisfied
    starts, ends = indptr[finl], indptr[finl + 1]
    cnt = ends - starts
    e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    np.subtract.at(deg, dst[e], 1)
    openm = ~fin
    frontier = np.flatnonzero((deg == 0) & openm)
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
    p, r = key // RB, key % RB
    sel = openm & (deg == 0)
    ntile = tile.max() + 1
    vkey = p[sel].astype(np.int64) * ntile + tile[sel]
    visits = np.unique(vkey).size
    rkey = vkey * RB + r[sel]
    u, c = np.unique(rkey, return_counts=True)
    wave_rounds = int(np.ceil(c / 64).sum())
    # per pass: visits, rounds of the longest visit
    pv = {}
    for k in np.unique(vkey) // ntile:
        pv[k] = pv.get(k, 0) + 1
    vmax = {}
    for k, rr in zip(u // RB // ntile, u % RB):
        vmax[k] = max(vmax.get(k, 0), rr)
    npass = int(p[sel].max()) if sel.any() else 0
    first = [(int(k), pv[k], vmax[k]) for k in sorted(pv)[:6]]
    return dict(open=int(openm.sum()), passes=npass, visits=visits, rounds=int(u.size), wave_rounds=wave_rounds,
                chain_sum=int(sum(vmax.values())), first=first)


def report(name, fin):
    s = tile_schedule(fin)
    print('%-34s open %9d (%5.2f%%)  passes %3d  visits %7d  rounds %8d  wave-rounds %8d  sum of longest visit per pass %5d  first passes (pass, visits, longest) %s'
          % (name, s['open'], 100.0 * s['open'] / NN, s['passes'], s['visits'], s['rounds'], s['wave_rounds'], s['chain_sum'], s['first']))
    sys.stdout.flush()


fin0 = np.zeros(NN, bool)
report('no prologue', fin0)

# ---- dense Kahn levels
deg = indeg.copy()
fin = np.zeros(NN, bool)
frontier = np.flatnonzero(deg == 0)
for k in range(1, 9):
    fin[frontier] = True
    starts, ends = indptr[frontier], indptr[frontier + 1]
    cnt = ends - starts
    e = np.repeat(starts, cnt) + (np.arange(cnt.sum()) - np.repeat(np.cumsum(cnt) - cnt, cnt))
    d = dst[e]
    np.subtract.at(deg, d, 1)
    frontier = np.unique(d[deg[d] == 0])
    if k in (3, 5, 8):
        report('dense levels: %d' % k, fin.copy())

# ---- directional sweeps
FLIP = np.array([5, 6, 7, 3, 4, 0, 1, 2])
TRANS = np.array([0, 3, 5, 1, 6, 2, 4, 7])


def remap(mask, table):
    out = np.zeros_like(mask)
    for d in range(8):
        out |= (((mask >> d) & 1) << table[d]).astype(np.uint8)
    return out


def sweep_south(fb, im, pit_ok, lanes, seg, iters):
    """one sweep marching +row over [rows, cols] arrays; returns the cells finished in this sweep"""
    R, Cn = fb.shape
    fs = np.zeros_like(fb)
    strip = np.arange(Cn) // lanes

    def shifted(a, dj):             # a[j + dj] with False outside, and only when the source column is in the same strip
        out = np.zeros(Cn, bool)
        if dj == 0:
            return a.copy()
        if dj == -1:
            out[1:] = a[:-1] & (strip[1:] == strip[:-1])
        else:
            out[:-1] = a[1:] & (strip[:-1] == strip[1:])
        return out

    def shifted_any(a, dj):         # a[j + dj] regardless of strips (final before the sweep: read from memory), True outside
        out = np.ones(Cn, bool)
        if dj == 0:
            return a.copy()
        if dj == -1:
            out[1:] = a[:-1]
        else:
            out[:-1] = a[1:]
        return out
    zero = np.zeros(Cn, bool); one = np.ones(Cn, bool)
    for r in range(R):
        vis_prev = r > 0 and (r // seg == (r - 1) // seg)
        okd = []
        for d in range(8):
            ddi, ddj = DI[d], DJ[d]
            rr = r + ddi
            if rr < 0 or rr >= R:
                okd.append(one); continue
            base = shifted_any(fb[rr], ddj)
            if ddi == -1 and vis_prev:
                base = base | shifted(fs[rr], ddj)
            okd.append(base)
        openr = ~fb[r]
        for it in range(iters):
            ready = openr & ~fs[r] & pit_ok[r]
            for d in range(8):
                need = ((im[r] >> d) & 1).astype(bool)
                ok = okd[d]
                if DI[d] == 0 and it > 0:
                    ok = ok | shifted(fs[r], DJ[d])
                ready &= ~need | ok
            if not ready.any():
                break
            fs[r] |= ready
    return fs


def dir_sweep(fin, direction):
    fb = fin.reshape(n, n)
    # pit in-edges: sources must be final before the sweep
    bad = np.bincount(pd[~fin[ps]], minlength=NN) > 0 if ps.size else np.zeros(NN, bool)
    pit_ok = (~bad).reshape(n, n)
    im = inmask
    if direction in 'NW':
        pass
    if direction == 'S':
        fs = sweep_south(fb, im, pit_ok, LANES, SEG, ITERS)
    elif direction == 'N':
        fs = sweep_south(fb[::-1], remap(im, FLIP)[::-1], pit_ok[::-1], LANES, SEG, ITERS)[::-1]
    elif direction == 'E':
        fs = sweep_south(fb.T, remap(im, TRANS).T, pit_ok.T, LANES, SEG, ITERS).T
    else:
        fs = sweep_south(fb.T[::-1], remap(remap(im, TRANS), FLIP).T[::-1], pit_ok.T[::-1], LANES, SEG, ITERS)[::-1].T
    return (fb | fs).reshape(-1).copy(), int(fs.sum())


for pattern in ('SNSNSNSN', 'SNEWSNEW'):
    fin = np.zeros(NN, bool)
    for k, dch in enumerate(pattern, 1):
        fin, got = dir_sweep(fin, dch)
        print('  sweep %d %s: +%d cells (%.2f%%), open %.2f%%' % (k, dch, got, 100.0 * got / NN, 100.0 * (NN - fin.sum()) / NN))
        if k in (2, 4, 6, 8):
            report('sweeps %s' % pattern[:k], fin.copy())
