#!/usr/bin/env python
"""debug: one case of tools/soak_conditioning_device.py, device vs host fill_flats, where do they differ"""
import os, sys, warnings
import numpy as np
from scipy import ndimage
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import soak_conditioning_device as S
from pydem_amd import DEMProcessor, conditioning
warnings.simplefilter('ignore')
k = int(sys.argv[1])
rec, z, o, dX, dY = S.make_case(k)
print(rec, 'nan cells', int(np.isnan(z).sum()))
want = conditioning.fill_flats(z, o['maximum_pit_area'], o['fill_flats_below_sea'], o['fill_flats_source_tol'], o['fill_flats_peaks'], o['fill_flats_pits'])
dp = DEMProcessor(elev=z.copy(), dX=dX, dY=dY, **o)
dp.calc_fill_flats()
print('on device:', 'elev' in dp._on_device)
got = np.array(dp.elev)
diff = ~((got == want) | (np.isnan(got) & np.isnan(want)))
print('differing cells', int(diff.sum()))
data = z if not o['maximum_pit_area'] else conditioning.fill_pit_artifacts(z, o['maximum_pit_area'], o['fill_flats_below_sea'])
data = np.asarray(data, float)
flat = (ndimage.minimum_filter(data, (3, 3)) >= data) & ((data != 0) if o['fill_flats_below_sea'] else (data > 0))
flat[0, 0] = flat[-1, 0] = flat[0, -1] = flat[-1, -1] = False
lab, nlab = ndimage.label(flat, structure=np.ones((3, 3), bool))
labs = np.unique(lab[diff])
print('host: flat cells', int(flat.sum()), 'regions', nlab)
print('labels of differing cells', labs[:20], 'cells outside any region:', int((lab[diff] == 0).sum()))
nanmask = np.isnan(data)
near = ndimage.binary_dilation(nanmask, np.ones((3, 3), bool))
for L in labs[:6]:
    if L == 0: continue
    reg = lab == L
    ii, jj = np.where(reg)
    ring = ndimage.binary_dilation(reg, np.ones((3, 3), bool)) & ~reg
    print('region', L, 'size', int(reg.sum()), 'bbox', ii.min(), ii.max(), jj.min(), jj.max(), 'level', data[reg][0], 'ring NaN cells', int(nanmask[ring].sum()),
          'ring == level', int((data[ring] == data[reg][0]).sum()), 'ring > level', int((data[ring] > data[reg][0]).sum()), 'ring < level', int((data[ring] < data[reg][0]).sum()),
          'on edge', bool((ii == 0).any() or (jj == 0).any() or (ii == z.shape[0] - 1).any() or (jj == z.shape[1] - 1).any()),
          'diff cells in it', int((diff & reg).sum()))
    d = np.where(diff & reg)
    for q in range(min(4, d[0].size)):
        i, j = d[0][q], d[1][q]
        print('   cell', i, j, 'in', data[i, j], 'host', want[i, j], 'device', got[i, j])
