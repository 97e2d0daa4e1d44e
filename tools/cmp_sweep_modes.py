#!/usr/bin/env python
"""GPU box: the compact passes (default) against the generic listed passes (PYDEM_SWEEP_COMPACT=0) on the same tiles:
masks must be identical, uca equal to rounding.   cmp_sweep_modes.py [size ...]"""
import os
import subprocess
import sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CHILD = r'''
import sys, warnings
import numpy as np
sys.path.insert(0, %(root)r)
warnings.simplefilter('ignore')
from pydem_amd import DEMProcessor, synth
n = %(n)d
z = synth.fractal(n, n, seed=1)
dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=%(pits)s)
dp.calc_slopes_directions(); dp.calc_uca()
np.savez(%(out)r, uca=dp.uca, todo=dp.edge_todo, done=dp.edge_done)
print('T', dp.timings)
'''
for n in [int(a) for a in sys.argv[1:]] or [1024, 4096]:
    for pits in (True, False):
        res = []
        for c in ('1', '0'):
            out = '/tmp/cmp_%s.npz' % c
            e = dict(os.environ, PYDEM_SWEEP_COMPACT=c)
            r = subprocess.run([sys.executable, '-c', CHILD % dict(root=ROOT, n=n, pits=pits, out=out)], env=e, capture_output=True, text=True)
            if r.returncode != 0:
                print('FAILED compact=%s n=%d pits=%s\n' % (c, n, pits), r.stdout[-1500:], r.stderr[-3000:]); sys.exit(1)
            res.append(dict(np.load(out)))
            print('n=%d pits=%s compact=%s %s' % (n, pits, c, [l for l in r.stdout.splitlines() if l.startswith('T')][-1][:400]))
        a, b = res
        nan_eq = np.array_equal(np.isnan(a['uca']), np.isnan(b['uca']))
        with np.errstate(invalid='ignore', divide='ignore'):
            rel = np.nanmax(np.abs(a['uca'] - b['uca']) / np.abs(b['uca']))
        print('   nan pattern equal %s, max rel diff %.3e, todo equal %s, done equal %s' % (
            nan_eq, rel, np.array_equal(a['todo'], b['todo']), np.array_equal(a['done'], b['done'])))
        if not (nan_eq and rel < 1e-11 and np.array_equal(a['todo'], b['todo']) and np.array_equal(a['done'], b['done'])):
            print('MISMATCH'); sys.exit(2)
print('CMP-OK')
