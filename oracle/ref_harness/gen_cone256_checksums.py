#!/usr/bin/env python
"""BASELINE.json config 1: the reference's own 256 x 256 synthetic cone (pydem/utils_test_pydem.py case 'cone', :98-103,
:422) through the UNMODIFIED reference DEMProcessor.calc_twi() with its defaults.  The arrays are too big to commit as a
fixture, so their sha256 (float64 bytes) go to tests/golden/cone256_reference.json; tests/test_oracle_golden.py holds the
oracle (and tests/test_gpu_parity.py the device path) against them.  Run through run.sh (imports the reference)."""
import hashlib
import json
import os
import sys
import warnings

from load_reference import load_reference

pydem = load_reference()
import numpy as np  # noqa: E402
from pydem.dem_processing import DEMProcessor  # noqa: E402
from pydem import utils_test_pydem as U  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, 'tests', 'golden', 'cone256_reference.json')


def sha(a, dt):
    return hashlib.sha256(np.ascontiguousarray(np.asarray(a), dt).tobytes()).hexdigest()


def main():
    nn = 256
    x, y = np.mgrid[-1:1:complex(0, nn), -1:1:complex(0, nn)]     # make_test_files (:422)
    raster, _, _ = U.case_cone(x, y)                              # test case 0 of the reference's utilities (:98-124)
    elev = np.ascontiguousarray(np.ma.filled(raster, np.nan), np.float64)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp = DEMProcessor(elev=elev.copy())
        twi = dp.calc_twi()
    rec = {'source': 'unmodified reference, DEMProcessor(elev=cone256).calc_twi() with default options',
           'shape': [nn, nn], 'elev_sha256': sha(elev, np.float64),
           'mag_sha256': sha(dp.mag, np.float64), 'direction_sha256': sha(dp.direction, np.float64),
           'flats_sha256': sha(dp.flats, np.uint8), 'uca_sha256': sha(dp.uca, np.float64), 'twi_sha256': sha(twi, np.float64),
           'edge_todo_sha256': sha(dp.edge_todo, np.uint8), 'edge_done_sha256': sha(dp.edge_done, np.uint8),
           'uca_max': float(np.nanmax(dp.uca)), 'uca_sum': float(np.nansum(dp.uca)), 'n_flats': int(np.asarray(dp.flats).sum()),
           'twi_nan': int(np.isnan(twi).sum())}
    json.dump(rec, open(OUT, 'w'), indent=1, sort_keys=True)
    print(rec)


if __name__ == '__main__':
    main()
