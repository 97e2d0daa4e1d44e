import os, sys, warnings, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.simplefilter('ignore')
import soak_pm as S
from pydem_amd import process_manager, synth
k = int(sys.argv[1])
rec, z, ny, nx, ov, dkw = S.make_case(k)
width = int(np.random.default_rng(77 + k + 1).choice([2, 3, 8]))
def run(**attrs):
    d = tempfile.mkdtemp()
    for t, (elev, bounds) in enumerate(synth.split_mosaic(z, ny, nx, ov)):
        np.savez(os.path.join(d, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
    process_manager.DEBUG = True
    pm = process_manager.ProcessManager(in_path=d, elev_conditioned=True, dem_proc_kwargs=dict(dkw), n_workers=width, edge_mode='pool')
    for a, v in attrs.items(): setattr(pm, a, v)
    pm.compute_grid(); pm.process_elevation(); pm.process_aspect_slope(); pm.process_uca()
    log = []
    orig = pm._run_edge_round_inner
    def traced(i, dp, data, done, todo, incremental=False):
        orig(i, dp, data, done, todo, incremental)
        n, m = pm.tiles_shape[i]
        lines = {}
        for nm in ('uca', 'edge_done', 'edge_todo'):
            for axis, idx in ((0, 0), (0, n - 1), (1, 0), (1, m - 1), (0, 1), (0, n - 2), (1, 1), (1, m - 2)):
                lines[(nm, axis, idx)] = np.array(pm.tiles[i].get_line(nm, axis, idx), float)
        log.append((pm.edge_waves, i, lines, {k2: np.array(v, float) for k2, v in data.items()}, {k2: np.array(v) for k2, v in done.items()}, {k2: np.array(v) for k2, v in todo.items()}))
    pm._run_edge_round_inner = traced
    pm.process_uca_edges()
    shutil.rmtree(d, ignore_errors=True)
    return log
A = run(edge_device_board=False, edge_incremental=False)
B = run(edge_device_board=False)
print('rounds', len(A), len(B))
for (wa, ia, la, da, dna, tda), (wb, ib, lb, db, dnb, tdb) in zip(A, B):
    assert (wa, ia) == (wb, ib), ((wa, ia), (wb, ib))
    for key in ('left', 'right', 'top', 'bottom'):
        for nm, x, y in (('data', da, db), ('done', dna, dnb), ('todo', tda, tdb)):
            xa, ya = np.asarray(x[key], float), np.asarray(y[key], float)
            if nm == 'data':
                xa = np.where(np.asarray(dna[key], bool), xa, 0); ya = np.where(np.asarray(dnb[key], bool), ya, 0)
            bad = ~np.isclose(xa, ya, rtol=1e-9, atol=1e-12, equal_nan=True)
            if bad.any():
                print('INPUT differs: wave', wa, 'tile', ia, key, nm, 'at', np.argwhere(bad).ravel()[:6].tolist(), xa[bad][:4], ya[bad][:4]); sys.exit(0)
    for kk in la:
        xa, xb = la[kk], lb[kk]
        if kk[0] == 'uca':                     # values only count where the cell is done (in both)
            dn = (la[('edge_done',) + kk[1:]] != 0) & (lb[('edge_done',) + kk[1:]] != 0)
            xa = np.where(dn, xa, 0); xb = np.where(dn, xb, 0)
        bad = ~np.isclose(xa, xb, rtol=1e-9, atol=1e-12, equal_nan=True)
        if bad.any():
            print('OUTPUT differs: wave', wa, 'tile', ia, kk, 'at', np.argwhere(bad).ravel()[:6].tolist(), 'plain', xa[bad][:4], 'inc', xb[bad][:4])
            idx = int(np.argwhere(bad).ravel()[0])
            for key in ('left', 'right', 'top', 'bottom'):
                L = len(da[key])
                print('   strip', key, 'len', L)
            sys.exit(0)
print('no difference on the edge lines')
