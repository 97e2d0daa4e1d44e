#!/usr/bin/env python
"""Golden vectors for the overview (block mean) step of the reference's export, calc_overview
(pydem/process_manager.py:317-352): a handful of arrays (sizes that divide by the factor and sizes that leave partial blocks
on the right / bottom / corner, NaN cells, an all-zero array) through the UNMODIFIED function, whole array = one chunk, with
the in-memory zarr stand-in.  Written to tests/golden/overview_cases.npz.  Run through run.sh."""
import os
import sys

from load_reference import load_reference

pydem = load_reference()
import numpy as np  # noqa: E402
import zarr  # noqa: E402  (the stand-in)
from pydem import process_manager  # noqa: E402

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(REPO, 'tests', 'golden', 'overview_cases.npz')


def main():
    rng = np.random.default_rng(5)
    cases = [((50, 71), 3), ((27, 27), 3), ((10, 100), 4), ((33, 8), 3), ((64, 65), 2), ((7, 7), 3)]
    rec = {}
    for k, (shape, factor) in enumerate(cases):
        data = rng.normal(size=shape) * 100 + 500
        if k == 2:
            data[3:5, 40:60] = np.nan
        if k == 4:
            data[:] = 0.0
        new_shape = [int(np.ceil(s / factor)) for s in shape]
        src = '/tmp/_ov_src_%d' % k
        dst = '/tmp/_ov_dst_%d' % k
        zarr.reset()
        a = zarr.open(src, mode='a', shape=shape, chunks=shape, dtype='float64')
        a[:] = data
        b = zarr.open(dst, mode='a', shape=new_shape, chunks=new_shape, dtype='float64')
        res = process_manager.calc_overview(src, dst, factor, (slice(0, shape[0]), slice(0, shape[1])),
                                            (slice(0, new_shape[0]), slice(0, new_shape[1])))
        rec['in_%d' % k] = data
        rec['factor_%d' % k] = np.int64(factor)
        rec['out_%d' % k] = np.array(b)
        rec['status_%d' % k] = np.int64(res[0])
        print(k, shape, factor, res[0], np.array(b).shape)
    rec['n_cases'] = np.int64(len(cases))
    np.savez_compressed(OUT, **rec)


if __name__ == '__main__':
    main()
