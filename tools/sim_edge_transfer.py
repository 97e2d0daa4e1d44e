#!/usr/bin/env python
"""Offline model (CPU, oracle graph) for the edge fix-up: how large is a tile's inlet -> outlet transfer operator?

The UCA correction a tile receives is linear in the values on its inlets: uca = own + sum_i T(c, i) * uca_in(i), with
T(c, i) the sum over flow paths from perimeter cell i to cell c of the products of the edge weights.  Restricted to
perimeter cells as outputs, T is what a fix-up round needs; this script counts its non-zeros (above a weight threshold)
and the lengths of the dependent paths the cascade walks today, on the bench terrain.   sim_edge_transfer.py [size]"""
import os
import sys
import numpy as np
import scipy.sparse as sp
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512
z = O.synth_fractal(n, n, seed=1)
o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
o.calc_slopes_directions(); o.build_graph()
indptr, indices, data = o.A            # CSC: column = from, rows = to
NN = n * n
A = sp.csc_matrix((data, indices, indptr), shape=(NN, NN)).tocsr()          # A[to, from]
ii, jj = np.divmod(np.arange(NN), n)
perim = np.flatnonzero((ii == 0) | (ii == n - 1) | (jj == 0) | (jj == n - 1))
P = perim.size
# X_k = A^k E (E = unit vectors of the perimeter cells); T = sum_k X_k restricted to perimeter rows
E = sp.csr_matrix((np.ones(P), (perim, np.arange(P))), shape=(NN, P))
X = E
is_perim = np.zeros(NN, bool); is_perim[perim] = True
T = sp.csr_matrix((P, P))
row_of = -np.ones(NN, np.int64); row_of[perim] = np.arange(P)
levels = 0
reach = np.zeros(P, np.int64)           # cells downstream of each inlet (with multiplicity of levels)
longest = np.zeros(P, np.int64)
while X.nnz:
    X = (A @ X).tocsr()
    levels += 1
    if X.nnz == 0:
        break
    cols = X.tocsc()
    nz_per_col = np.diff(cols.indptr)
    reach += nz_per_col
    longest[nz_per_col > 0] = levels
    sel = X[perim]
    T = T + sel
T = T.tocsc()
print('tile %d x %d: perimeter %d cells, graph %d edges, cascade depth from the perimeter %d levels' % (n, n, P, A.nnz, levels))
print('dependent levels per inlet: median %d, p90 %d, max %d' % (np.median(longest), np.percentile(longest, 90), longest.max()))
print('cells below an inlet (sum over levels): median %d, p90 %d, max %d' % (np.median(reach), np.percentile(reach, 90), reach.max()))
for thr in (0.0, 1e-12, 1e-9, 1e-6):
    m = np.abs(T.data) > thr
    per_in = np.bincount(np.repeat(np.arange(P), np.diff(T.indptr))[m], minlength=P)
    print('transfer entries with weight > %g: %d (%.1f per inlet, max %d; %.2f per perimeter cell pair in 1e-3)'
          % (thr, m.sum(), m.sum() / P, per_in.max(), 1e3 * m.sum() / (P * P)))
