"""The BASELINE.json configurations at their stated sizes, on the device, against checksums of the CPU oracle that
were generated on the build box by tools/gen_large_checksums.py (tests/golden/large_checksums.json):

  config 3   16384 x 16384 fp64 fractal tile (the bench tile), full path: section / flats / pit -> drain pairs /
             edge masks bit for bit (sha256), mag / direction / uca / twi through NaN counts, extrema, sums and
             quantiles at 1e-6 -- plus invariants that need no oracle (uca >= cell area, NaN <=> flats)
  config 5   8192 x 8192 int16 SRTM-like tile with the reference's defaults: the conditioned surfaces (flats filled,
             pit paths carved) bit for bit, then the same fields
  config 4   an 8-tile 2 x 4 mosaic (one-pixel overlap, the bench layout) at 512^2 per tile: serial and pool edge
             schedules against the same host logic with the oracle-backed processor

The 1024^2 entries of the checksum file run first (seconds); the full sizes need ~25 GB of HBM and a minute.
"""
import hashlib
import json
import os
import warnings

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
SUMS = json.load(open(os.path.join(ROOT, 'tests', 'golden', 'large_checksums.json')))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def check_float(name, got, want, rtol=1e-6):
    got = np.asarray(got, np.float64)
    assert int(np.isnan(got).sum()) == want['nan'], name + ': NaN count'
    assert int(np.isneginf(got).sum()) == want['neg_inf'] and int(np.isposinf(got).sum()) == want['pos_inf'], name + ': infinities'
    v = got[np.isfinite(got)]
    assert np.isclose(v.min(), want['min'], rtol=rtol, atol=1e-12), (name, 'min', v.min(), want['min'])
    assert np.isclose(v.max(), want['max'], rtol=rtol, atol=1e-12), (name, 'max', v.max(), want['max'])
    assert np.isclose(v.sum(), want['sum'], rtol=rtol), (name, 'sum', v.sum(), want['sum'])
    q = np.quantile(v, SUMS[next(iter(SUMS))]['quantile_levels'])
    assert np.allclose(q, want['quantiles'], rtol=rtol, atol=1e-9), (name, 'quantiles', q, want['quantiles'])


def check_fields(dp, want, cell_area):
    assert sha(np.asarray(dp.section, np.int8)) == want['section_sha256'], 'section'
    flats = np.asarray(dp.flats, bool)
    assert int(flats.sum()) == want['n_flats'] and sha(flats.astype(np.uint8)) == want['flats_sha256'], 'flats'
    src, dst, _ = dp._tile.pit_edges()
    keep = src >= 0
    pairs = np.stack([src[keep].astype(np.int64), dst[keep].astype(np.int64)], 1)
    pairs = pairs[np.lexsort((pairs[:, 1], pairs[:, 0]))]
    assert pairs.shape[0] == want['n_pit_pairs'] and sha(pairs) == want['pit_pairs_sha256'], 'pit -> drain assignments'
    assert sha(np.asarray(dp.edge_todo, np.uint8)) == want['edge_todo_sha256'], 'edge_todo'
    assert sha(np.asarray(dp.edge_done, np.uint8)) == want['edge_done_sha256'], 'edge_done'
    uca = dp.uca
    check_float('uca', uca, want['uca'])
    # invariants: every cell holds at least its own area; NaN exactly on the flats that are left
    assert np.array_equal(np.isnan(uca), flats)
    assert np.nanmin(uca) >= cell_area * (1 - 1e-12)
    del uca
    check_float('mag', dp.mag, want['mag'])
    check_float('direction', dp.direction, want['direction'])
    check_float('twi', dp.twi, want['twi'])


@pytest.mark.parametrize('size', [1024, 16384])
def test_config3_full_path_against_oracle_checksums(size):
    key = 'config3_%d' % size
    if key not in SUMS:
        pytest.skip("no checksums for %s (tools/gen_large_checksums.py)" % key)
    from pydem_amd import DEMProcessor
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp = DEMProcessor.from_synthetic((size, size), dict(seed=1), dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False,
                                         drain_pits=True)
        dp.run_twi()
    check_fields(dp, SUMS[key], 900.0)
    if size == 16384:
        # the pit pairs above came through every pass of the search: the row pass (16 lanes per pit) ran by its default rule
        # (pits beyond the lane pass outnumber the resident wavefronts) and left only a small part to the wavefront pass
        tm = dp.timings
        assert tm['n_pits_row'] > 100000 and 0 < tm['n_pits_wave'] < tm['n_pits_row'] // 20, tm


@pytest.mark.parametrize('size', [1024, 8192])
def test_config5_int16_defaults_against_oracle_checksums(size):
    key = 'config5_%d' % size
    if key not in SUMS:
        pytest.skip("no checksums for %s (tools/gen_large_checksums.py)" % key)
    from pydem_amd import DEMProcessor, synth
    want = SUMS[key]
    z = synth.srtm_int16(size, size, seed=3)
    assert sha(z) == want['input_sha256']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0)            # reference defaults
        dp.calc_fill_flats()
        assert 'elev' in dp._on_device, "the flats step did not run on the device"
        assert sha(np.asarray(dp.elev, np.float64)) == want['filled_sha256'], 'surface after calc_fill_flats'
        res = dp._pit_paths_on_device()
        assert res is not None, "the pit drain paths fell back to the host loop"
        assert (res[0], res[1]) == (want['paths_failed'], want['paths_iterations'])
        assert sha(np.asarray(dp.elev, np.float64)) == want['drained_sha256'], 'surface after calc_pit_drain_paths'
        dp.fill_flats = False
        dp.drain_pits_path = False
        dp.run_twi()
    check_fields(dp, want, 900.0)


@pytest.mark.parametrize('n_workers', [1, 8])
def test_config4_mosaic_against_oracle_backed_flow(n_workers, tmp_path):
    """2 x 4 tiles of 512 x 512, one-pixel overlap, fractal with pits -- the bench's layout (bench.tile_specs) through the
    ProcessManager: reference schedule and pool waves, device processor against the oracle-backed processor."""
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from oracle import oracle as O
    from oracle_processor import OracleProcessor
    from pydem_amd import process_manager
    n = 512
    specs = bench.tile_specs(8, n, n)
    host_specs = []
    for sp in specs:
        sp2 = dict(sp)
        sy = sp2.pop('synth')
        sp2['elev'] = O.synth_fractal(n, n, seed=sy['seed'], row0=sy['row0'], col0=sy['col0'])
        host_specs.append(sp2)
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pm = process_manager.ProcessManager(elev_source_files=specs, elev_conditioned=True, dem_proc_kwargs={'drain_pits': True},
                                            n_workers=n_workers, tiles_in_flight=2)
        pm.process_twi()
        po = process_manager.ProcessManager(elev_source_files=host_specs, elev_conditioned=True, dem_proc_kwargs={'drain_pits': True},
                                            n_workers=n_workers, processor_cls=OracleProcessor)
        po.process_twi()
    assert (pm.edge_rounds, pm.edge_waves) == (po.edge_rounds, po.edge_waves)
    if n_workers == 8:
        assert pm.edge_queued_batches > 0           # (8 tiles <= 2 * n_workers: the waves were chosen on the device and queued)
    for i in range(8):
        assert np.array_equal(pm.tile_result(i, 'edge_todo'), po.tile_result(i, 'edge_todo')), i
        assert np.array_equal(pm.tile_result(i, 'edge_done'), po.tile_result(i, 'edge_done')), i
        a, b = pm.tile_result(i, 'uca_total'), po.tile_result(i, 'uca_total')
        assert np.array_equal(np.isnan(a), np.isnan(b)), i
        assert np.allclose(a, b, rtol=1e-9, atol=0, equal_nan=True), (i, np.nanmax(np.abs(a - b) / np.abs(b)))
        ta, tb = pm.tile_result(i, 'twi'), po.tile_result(i, 'twi')
        fin = np.isfinite(ta) & np.isfinite(tb)
        assert np.allclose(ta[fin], tb[fin], rtol=1e-9, atol=1e-9), i


@pytest.mark.parametrize('tile', [2048, 8192])
def test_config4_mosaic_2048_against_oracle_checksums(tile):
    """The directory path above toy sizes: 2 x 4 tiles of 2048 x 2048 -- and of 8192 x 8192, BASELINE config 4 as stated, about
    50 GB resident on the one GPU -- (bench layout, one-pixel overlap, pits), pool schedule of width 8 with the device edge
    board, against checksums of the same flow with the oracle-backed processor (tools/gen_large_checksums.py 4 [size=8192]:
    5 / 75 minutes of oracle time on the build box).  Same waves and rounds, edge masks bit for bit, uca_total / twi per tile
    through NaN counts, extrema, sums and quantiles.  Reference: pydem/process_manager.py:224-284, 1090-1246 (the
    multi-worker schedule as deterministic waves)."""
    key = 'config4_8x%d' % tile
    if key not in SUMS:
        pytest.skip("no checksums for %s (tools/gen_large_checksums.py 4 size=%d)" % (key, tile))
    import sys
    sys.path.insert(0, ROOT)
    import bench
    from pydem_amd import process_manager
    want = SUMS[key]
    n = want['tile']
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pm = process_manager.ProcessManager(elev_source_files=bench.tile_specs(8, n, n), elev_conditioned=True,
                                            dem_proc_kwargs={'drain_pits': True}, n_workers=want['n_workers'])
        pm.process_twi()
    assert (pm.edge_rounds, pm.edge_waves) == (want['edge_rounds'], want['edge_waves'])
    assert pm.edge_queued_batches > 0               # (pool width 8: queued waves, device-built operators)
    for i, w in enumerate(want['tiles']):
        assert sha(np.asarray(pm.tile_result(i, 'edge_todo'), np.uint8)) == w['edge_todo_sha256'], (i, 'edge_todo')
        assert sha(np.asarray(pm.tile_result(i, 'edge_done'), np.uint8)) == w['edge_done_sha256'], (i, 'edge_done')
        check_float('tile %d uca_total' % i, pm.tile_result(i, 'uca_total'), w['uca_total'])
        check_float('tile %d twi' % i, pm.tile_result(i, 'twi'), w['twi'])
    # cell by cell where an error of the fix-up (condensed rounds re-associate sums; strips travel between tiles) would live: the
    # two outermost lines of every side of every tile against the oracle-backed flow (masks exactly, uca_total to 1e-9)
    strips_fn = os.path.join(ROOT, 'tests', 'golden', 'config4_strips_8x%d.npz' % tile)
    if os.path.exists(strips_fn):
        S = np.load(strips_fn)
        for i in range(8):
            for name in ('uca_total', 'edge_todo', 'edge_done'):
                a = np.asarray(pm.tile_result(i, name))
                for side, line in (('r0', a[0]), ('r1', a[1]), ('rm2', a[-2]), ('rm1', a[-1]), ('c0', a[:, 0]), ('c1', a[:, 1]), ('cm2', a[:, -2]), ('cm1', a[:, -1])):
                    want_line = S['t%d_%s_%s' % (i, name, side)]
                    if name == 'uca_total':
                        assert np.array_equal(np.isnan(line), np.isnan(want_line)), (i, name, side)
                        assert np.allclose(line, want_line, rtol=1e-9, atol=0, equal_nan=True), (i, name, side, np.nanmax(np.abs(line - want_line) / np.abs(want_line)))
                    else:
                        assert np.array_equal(np.asarray(line, np.uint8), want_line), (i, name, side)
    else:
        assert tile != 2048, "tests/golden/config4_strips_8x2048.npz is missing (tools/gen_large_checksums.py 4)" 
