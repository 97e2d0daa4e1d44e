#!/usr/bin/env python
"""One case of tools/soak_parity.py in detail: where the device and the oracle differ.   dbg_soak_case.py <case>"""
import os, sys, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import soak_parity as S
from oracle import oracle as O
from pydem_amd import DEMProcessor
k = int(sys.argv[1])
rec, z, kw, opt = S.make_case(k)
print(rec, {a: (b if np.isscalar(b) else 'array') for a, b in kw.items()})
with warnings.catch_warnings():
    warnings.simplefilter('ignore')
    o = O.OracleDEM(z, **kw, **opt); o.calc_twi()
    dp = DEMProcessor(elev=z, fill_flats=False, drain_pits_path=False, **kw, **opt); dp.calc_twi()
for nm in ('section', 'flats'):
    a, b = np.asarray(getattr(dp, nm)), np.asarray(getattr(o, nm))
    bad = np.argwhere(a != b)
    print(nm, 'differs at', len(bad), 'cells')
    for i, j in bad[:8]:
        print('  cell', (i, j), 'device', a[i, j], 'oracle', b[i, j], 'direction device %r oracle %r' % (dp.direction[i, j], o.direction[i, j]),
              'mag %r / %r' % (dp.mag[i, j], o.mag[i, j]))
        print('  window', z[max(i - 1, 0):i + 2, max(j - 1, 0):j + 2].tolist())
d = np.abs(dp.direction - o.direction)
print('direction: max abs diff', np.nanmax(d), 'cells with any diff', int((d > 0).sum()))
