#!/usr/bin/env python
"""Checksums of the CPU oracle at the BASELINE.json sizes that are too big to commit as arrays:

    config 3   16384 x 16384 fp64 fractal tile, seed 1 (the bench tile), drain_pits=True, no conditioning
    config 5   8192 x 8192 int16 SRTM-like tile, seed 3, reference defaults (fill_flats, drain_pits_path, drain_pits):
               conditioning by the host implementation (pydem_amd/conditioning.py, pinned bit for bit by the reference's
               g5_* / g7_* goldens), everything after it by the oracle

    config 4   8-tile 2 x 4 mosaic (one-pixel overlap, the bench layout) at 2048^2 per tile, pool edge schedule of width 8
               through the ProcessManager with the oracle-backed processor: per tile the edge masks, uca_total and twi after
               the fix-up, plus the round / wave counts of the schedule (4-5 minutes of oracle time)

Exact fields (section, flats, pit -> drain pairs, the conditioned surface) are recorded as sha256, float fields as NaN
count / min / max / pairwise sum / quantiles.  tests/test_gpu_large_configs.py compares the device results with them.
Run on the build box (oracle only -- the reference is not needed):  python tools/gen_large_checksums.py [3] [4] [5] [small] [size=8192]
(config 4 at its BASELINE size, 8 x 8192^2: `4 size=8192`, about 75 minutes and 50 GB)
"""
import hashlib
import json
import os
import sys
import time
import warnings

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle as O                      # noqa: E402
from pydem_amd import synth                          # noqa: E402

OUT = os.path.join(ROOT, 'tests', 'golden', 'large_checksums.json')
QS = [0.001, 0.01, 0.1, 0.25, 0.5, 0.75, 0.9, 0.99, 0.999]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def fstats(a):
    a = np.asarray(a, np.float64)
    ok = np.isfinite(a)
    v = a[ok]
    return {'nan': int(np.isnan(a).sum()), 'neg_inf': int(np.isneginf(a).sum()), 'pos_inf': int(np.isposinf(a).sum()),
            'min': float(v.min()), 'max': float(v.max()), 'sum': float(v.sum()), 'quantiles': [float(x) for x in np.quantile(v, QS)]}


def summarize(o, extra=None):
    pairs = np.stack([np.asarray(o.pit_i, np.int64), np.asarray(o.pit_j, np.int64)], 1) if o.pit_i is not None and len(o.pit_i) else np.zeros((0, 2), np.int64)
    pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    out = {'shape': list(o.elev.shape), 'section_sha256': sha(np.asarray(o.section, np.int8)), 'flats_sha256': sha(np.asarray(o.flats, np.uint8)),
           'pit_pairs_sha256': sha(pairs), 'n_pit_pairs': int(pairs.shape[0]), 'n_flats': int(np.asarray(o.flats).sum()),
           'edge_todo_sha256': sha(np.asarray(o.edge_todo, np.uint8)), 'edge_done_sha256': sha(np.asarray(o.edge_done, np.uint8)),
           'mag': fstats(o.mag), 'direction': fstats(o.direction), 'uca': fstats(o.uca), 'twi': fstats(o.twi), 'quantile_levels': QS}
    out.update(extra or {})
    return out


def config3(size=16384):
    z = O.synth_fractal(size, size, seed=1)
    t0 = time.time()
    o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True)
    o.calc_twi()
    return summarize(o, {'what': 'fp64 fractal seed 1, dX=dY=30, drain_pits=True, no conditioning', 'oracle_seconds': time.time() - t0})


def config5(size=8192):
    from pydem_amd import conditioning as C
    z = synth.srtm_int16(size, size, seed=3)
    dX = np.full(size - 1, 30.0); dY = np.full(size - 1, 30.0)
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        filled = C.fill_flats(z)
        t1 = time.time()
        drained, n_failed, used = C.pit_drain_paths(filled.copy(), dX, dY)
        t2 = time.time()
        o = O.OracleDEM(drained, dX=30.0, dY=30.0, drain_pits=True)
        o.calc_twi()
    return summarize(o, {'what': 'int16 SRTM-like seed 3, reference defaults', 'input_sha256': sha(z), 'filled_sha256': sha(filled),
                         'drained_sha256': sha(drained), 'paths_failed': int(n_failed), 'paths_iterations': int(used),
                         'host_fill_flats_seconds': t1 - t0, 'host_pit_paths_seconds': t2 - t1, 'oracle_seconds': time.time() - t2})


def config4(size=2048, n_workers=8):
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import bench
    from oracle_processor import OracleProcessor
    from pydem_amd import process_manager
    specs = []
    for sp in bench.tile_specs(8, size, size):
        sp2 = dict(sp)
        sy = sp2.pop('synth')
        sp2['elev'] = O.synth_fractal(size, size, seed=sy['seed'], row0=sy['row0'], col0=sy['col0'])
        specs.append(sp2)
    t0 = time.time()
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        po = process_manager.ProcessManager(elev_source_files=specs, elev_conditioned=True, dem_proc_kwargs={'drain_pits': True},
                                            n_workers=n_workers, processor_cls=OracleProcessor)
        po.process_twi()
    tiles = []
    # cell-level pin of the fix-up where its errors would live: the two outermost lines of every side of every tile (uca_total and
    # both edge masks), tests/golden/config4_strips_8x<size>.npz -- sums and quantiles cannot see a handful of wrong cells
    strips = {}
    for i in range(8):
        for name, dt in (('uca_total', np.float64), ('edge_todo', np.uint8), ('edge_done', np.uint8)):
            a = np.asarray(po.tile_result(i, name), dt)
            for side, line in (('r0', a[0]), ('r1', a[1]), ('rm2', a[-2]), ('rm1', a[-1]), ('c0', a[:, 0]), ('c1', a[:, 1]), ('cm2', a[:, -2]), ('cm1', a[:, -1])):
                strips['t%d_%s_%s' % (i, name, side)] = np.ascontiguousarray(line)
    np.savez_compressed(os.path.join(ROOT, 'tests', 'golden', 'config4_strips_8x%d.npz' % size), **strips)
    for i in range(8):
        twi = np.asarray(po.tile_result(i, 'twi'), np.float64)
        tiles.append({'edge_todo_sha256': sha(np.asarray(po.tile_result(i, 'edge_todo'), np.uint8)),
                      'edge_done_sha256': sha(np.asarray(po.tile_result(i, 'edge_done'), np.uint8)),
                      'uca_total': fstats(po.tile_result(i, 'uca_total')), 'twi': fstats(twi)})
    return {'what': '2 x 4 mosaic of %d^2 fp64 fractal tiles (bench.tile_specs), drain_pits=True, pool schedule n_workers=%d, oracle-backed processor'
                    % (size, n_workers), 'tile': size, 'n_workers': n_workers, 'edge_rounds': int(po.edge_rounds), 'edge_waves': int(po.edge_waves),
            'tiles': tiles, 'quantile_levels': QS, 'oracle_seconds': time.time() - t0}


def main():
    which = sys.argv[1:] or ['3', '5']
    small = 'small' in which
    tile4 = [int(w.split('=')[1]) for w in which if w.startswith('size=')]      # config 4 at another tile size: 4 size=8192
    res = json.load(open(OUT)) if os.path.exists(OUT) else {}
    if '3' in which:
        key = 'config3_%d' % (1024 if small else 16384)
        res[key] = config3(1024 if small else 16384)
        print(key, 'done in %.0f s' % res[key]['oracle_seconds'], flush=True)
        json.dump(res, open(OUT, 'w'), indent=1, sort_keys=True)
    if '4' in which:
        size4 = tile4[0] if tile4 else (512 if small else 2048)
        key = 'config4_8x%d' % size4
        r4 = config4(size4)
        res = json.load(open(OUT)) if os.path.exists(OUT) else {}       # (hours may have passed: re-read before writing)
        res[key] = r4
        print(key, 'done in %.0f s' % res[key]['oracle_seconds'], flush=True)
        json.dump(res, open(OUT, 'w'), indent=1, sort_keys=True)
    if '5' in which:
        key = 'config5_%d' % (1024 if small else 8192)
        res[key] = config5(1024 if small else 8192)
        print(key, 'done', flush=True)
        json.dump(res, open(OUT, 'w'), indent=1, sort_keys=True)


if __name__ == '__main__':
    main()
