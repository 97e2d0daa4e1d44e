import os, sys, warnings, tempfile, shutil
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, 'tools')); sys.path.insert(0, os.path.join(ROOT, 'tests'))
warnings.simplefilter('ignore')
import soak_pm as S
from pydem_amd import process_manager, synth
from oracle_processor import OracleProcessor
k = int(sys.argv[1]); ti = int(sys.argv[2]); r = int(sys.argv[3]); c = int(sys.argv[4])
rec, z, ny, nx, ov, dkw = S.make_case(k)
width = int(np.random.default_rng(77 + k + 1).choice([2, 3, 8])); print('width', width)
def run(cls, **attrs):
    d = tempfile.mkdtemp()
    for t, (elev, bounds) in enumerate(synth.split_mosaic(z, ny, nx, ov)):
        np.savez(os.path.join(d, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
    process_manager.DEBUG = True
    kw = dict(processor_cls=cls) if cls is not None else {}
    pm = process_manager.ProcessManager(in_path=d, elev_conditioned=True, dem_proc_kwargs=dict(dkw), n_workers=width, edge_mode='pool', **kw)
    for a, v in attrs.items(): setattr(pm, a, v)
    pm.compute_grid(); pm.process_elevation(); pm.process_aspect_slope(); pm.process_uca()
    u0 = float(np.asarray(pm.tiles[ti].uca)[r, c]); f0 = int(np.asarray(pm.tiles[ti].flats)[r, c]); d0 = int(np.asarray(pm.tiles[ti].edge_done)[r, c]); t0 = int(np.asarray(pm.tiles[ti].edge_todo)[r, c])
    if attrs.get('trace'):
        orig = pm._run_edge_round_inner
        def traced(i, dp, data, done, todo, incremental=False):
            if i == ti:
                key = 'right' if c == pm.tiles_shape[i][1] - 1 else ('left' if c == 0 else ('bottom' if r == pm.tiles_shape[i][0] - 1 else 'top'))
                idx = r if key in ('left', 'right') else c
                print('   round of tile', i, 'wave', pm.edge_waves, 'inc', incremental, key, 'data', data[key][idx], 'done', done[key][idx], 'todo', todo[key][idx])
            orig(i, dp, data, done, todo, incremental)
            if i == ti:
                line = pm.tiles[i].get_line('uca', 1, c)
                print('      -> uca', line[r], 'edge_done', pm.tiles[i].get_line('edge_done', 1, c)[r], 'edge_todo', pm.tiles[i].get_line('edge_todo', 1, c)[r])
        pm._run_edge_round_inner = traced
    pm.process_uca_edges()
    u1 = float(np.asarray(pm.tiles[ti].uca)[r, c])
    shutil.rmtree(d, ignore_errors=True)
    return u0, f0, d0, t0, u1, pm.edge_rounds, pm.edge_waves
print('oracle            ', run(OracleProcessor))
print('device board+inc  ', run(None))
print('device numpy+inc  ', run(None, edge_device_board=False, trace=True))
print('device numpy+plain', run(None, edge_device_board=False, edge_incremental=False, trace=True))
sys.exit(0)
print('device numpy+inc  ', run(None, edge_device_board=False, trace=True))
print('device numpy+plain', run(None, edge_device_board=False, edge_incremental=False, trace=True))
