"""How fast is the CPU oracle (the "port" bench.py times as `cpu_baseline`) relative to the UNMODIFIED reference on the
bench workload?  Build container only (needs /root/reference): the reference's DEMProcessor and the oracle run the same
fp64 fractal tiles (seed 1, dX = dY = 30 m, fill_flats=False, drain_pits_path=False -- the bench configuration), with
drain_pits on and off, single thread each; results must agree (uca) and the ratio port / reference is written to
profiles/r02_cpu_calibration.json, from which bench.py derives `cpu_baseline.reference_equiv`.

    bash oracle/ref_harness/run.sh oracle/ref_harness/calibrate_cpu_baseline.py [sizes...]
"""
import json
import os
import sys
import time
import warnings

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
from load_reference import load_reference      # noqa: E402  (before numpy)
pydem = load_reference()
import numpy as np                              # noqa: E402
sys.path.insert(0, ROOT)
from oracle import oracle as O                  # noqa: E402
from pydem.dem_processing import DEMProcessor  # noqa: E402  (the reference's)


def run(size, pits):
    z = O.synth_fractal(size, size, seed=1)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        t0 = time.perf_counter()
        # scalar spacings: the reference derives its on-grid dX2 / dY2 (the cell areas) only from scalars (:233-242)
        ref = DEMProcessor(elev=z.copy(), dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=pits)
        ref.calc_twi()
        t1 = time.perf_counter()
        o = O.OracleDEM(z.copy(), dX=30.0, dY=30.0, drain_pits=pits)
        o.calc_twi()
        t2 = time.perf_counter()
    same = bool(np.array_equal(np.asarray(ref.uca), o.uca, equal_nan=True) and np.array_equal(np.asarray(ref.twi), o.twi, equal_nan=True))
    cells = size * size / 1e6
    return {'size': size, 'drain_pits': bool(pits), 'reference_s': t1 - t0, 'port_s': t2 - t1, 'reference_Mcells_s': cells / (t1 - t0),
            'port_Mcells_s': cells / (t2 - t1), 'port_over_reference': (t1 - t0) / (t2 - t1), 'uca_twi_bit_identical': same}


def main():
    sizes = [int(a) for a in sys.argv[1:]] or [512, 1024, 2048]
    rows = []
    for size in sizes:
        for pits in (True, False):
            rows.append(run(size, pits))
            print(rows[-1], flush=True)
    bench_rows = [r for r in rows if r['drain_pits']]
    out = {'what': 'unmodified reference (DEMProcessor.calc_twi, one thread) vs the CPU oracle on the bench workload '
                   '(fp64 fractal seed 1, dX=dY=30, fill_flats=False, drain_pits_path=False)',
           'host': 'build container, %d cores' % os.cpu_count(), 'rows': rows,
           # bench.py uses the ratio at the largest size measured with drain_pits=True (the bench setting)
           'port_over_reference_bench': bench_rows[-1]['port_over_reference'],
           'port_over_reference_range_pits': [min(r['port_over_reference'] for r in bench_rows), max(r['port_over_reference'] for r in bench_rows)]}
    with open(os.path.join(ROOT, 'profiles', 'r02_cpu_calibration.json'), 'w') as f:
        json.dump(out, f, indent=1)
    print('port / reference on the bench workload:', out['port_over_reference_range_pits'])


if __name__ == '__main__':
    main()
