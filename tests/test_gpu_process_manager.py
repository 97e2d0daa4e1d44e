"""GPU: the ProcessManager drop-in on multi-tile mosaics against per-tile results of the unmodified
reference ProcessManager (tests/golden/pm_*.npz; n_workers=1, DEBUG spacing like the reference's own
tests).  Tiles start from the reference's conditioned elevation (the conditioning kernels are held to the
reference in tests/test_gpu_conditioning.py).  The edge fix-up follows the reference's serial visiting order; the sums inside a round differ in order, so
float fields are compared at 1e-9 relative; masks, NaN patterns and facet-derived fields exactly.
Pool mode (the multi-worker schedule): the reference's acceptance cases and the oracle-backed flow."""
import os

import numpy as np
import pytest

from conftest import golden_names, load_golden

pytestmark = pytest.mark.gpu


def _close(a, b, what, rtol=1e-9, atol=1e-12):
    a = np.asarray(a, float); b = np.asarray(b, float)
    assert a.shape == b.shape, what
    assert np.array_equal(np.isnan(a), np.isnan(b)), "%s: NaN pattern differs (%d vs %d NaN)" % (what, np.isnan(a).sum(), np.isnan(b).sum())
    ok = np.isclose(a, b, rtol=rtol, atol=atol, equal_nan=True)
    assert ok.all(), "%s: %d cells differ, worst rel %g" % (what, (~ok).sum(), np.nanmax(np.abs(a - b) / np.maximum(np.abs(b), 1e-300)))


@pytest.mark.parametrize('name', golden_names('pm_'))
def test_directory_flow_matches_reference(name, tmp_path):
    from test_process_manager_cpu import compare_with_golden, run_pm
    g = load_golden(name)
    pm, compact, order = run_pm(g, str(tmp_path))
    compare_with_golden(pm, compact, order, g, _close)
    assert pm.edge_rounds >= 1


@pytest.mark.parametrize('name', ['pm_fractal_2x3_ov1', 'pm_int16_defaults_2x2_ov2'])
def test_directory_flow_from_geotiff_tiles_on_the_device(name, tmp_path):
    """Real input format: the mosaic's tiles as (projected, Deflate, 16 x 16-tiled) GeoTIFF files read by pydem_amd/raster.py
    (the reference opens them with rasterio, pydem/utils.py:43-51) and fed to the device flow -- the same per-tile and stitched
    results as the reference's run on these tiles."""
    from pydem_amd import process_manager, raster
    from test_process_manager_cpu import compare_with_golden
    g = load_golden(name)
    for i in range(int(g['n_tiles'])):
        elev = g['t%02d_elev' % i]
        left, bottom, right, top = [float(v) for v in g['t%02d_bounds' % i]]
        n, m = elev.shape
        raster.write_geotiff(str(tmp_path / ('tile_%03d.tif' % i)), elev, ((right - left) / m, 0.0, left, 0.0, -(top - bottom) / n, top),
                             projected=True, compress=True, tile=16)
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}
    process_manager.DEBUG = True
    try:
        pm = process_manager.ProcessManager(in_path=str(tmp_path), dem_proc_kwargs=dkw, elev_conditioned=True)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pm.process_twi()
            compact = pm.save_non_overlap_data()
    finally:
        process_manager.DEBUG = False
    order = [int(np.argmin([np.abs(g['t%02d_bounds' % j] - pm.index[i, :4]).sum() for j in range(pm.n_inputs)])) for i in range(pm.n_inputs)]
    compare_with_golden(pm, compact, order, g, _close)
    out = str(tmp_path / 'twi.tif')
    pm.save_geotiff(out, 'twi', 'float32', overview_type='average')
    assert np.array_equal(raster.read_geotiff(out).array, compact['twi'].astype('float32'), equal_nan=True)


@pytest.mark.parametrize('name', ['pm_fractal_2x3_ov1', 'pm_cone32_4x5_ov3'])
def test_directory_flow_with_tiles_in_flight(name, tmp_path):
    """Several tiles of one GPU worked on at once (worker threads, one HIP stream per tile): same results."""
    from test_process_manager_cpu import compare_with_golden, run_pm
    g = load_golden(name)
    pm, compact, order = run_pm(g, str(tmp_path), tiles_in_flight=3)
    compare_with_golden(pm, compact, order, g, _close)


def test_multi_tile_equals_single_tile_on_cone(tmp_path):
    """The reference's own acceptance test (pydem/test/test_end_to_end.py:86-149): on the pit-free cone
    the stitched multi-tile UCA equals the single-tile UCA on [1:-1, 1:-1] to 6 decimals."""
    import os
    from pydem_amd import DEMProcessor, process_manager, synth
    nn = 96
    cone = synth.cone_scaled(nn)
    single = DEMProcessor(elev=cone, fill_flats=False, drain_pits_path=False)
    single.calc_twi()
    for k, (tiles, ov) in enumerate([((3, 3), 2), ((3, 3), 1), ((4, 3), 3)]):
        d = str(tmp_path / ('case%d' % k))
        os.makedirs(d)
        for t, (elev, bounds) in enumerate(synth.split_mosaic(cone, tiles[0], tiles[1], ov)):
            np.savez(os.path.join(d, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
        process_manager.DEBUG = True
        try:
            pm = process_manager.ProcessManager(in_path=d, elev_conditioned=True)
            pm.process_twi()
            compact = pm.save_non_overlap_data()
        finally:
            process_manager.DEBUG = False
        np.testing.assert_array_almost_equal(compact['uca'][1:-1, 1:-1], single.uca[1:-1, 1:-1], decimal=6)


@pytest.mark.parametrize('n_workers', [1, 8])
def test_rccl_transport_single_rank(n_workers, tmp_path):
    """The RCCL strip transport end to end on one GPU (world size 1: the collective still runs):
    device pack -> ncclAllReduce -> host views must reproduce the in-process transport's result (n_workers=1: the
    serial loop against the reference golden; n_workers=8: pool waves on the device edge board against the same
    waves with in-process strips)."""
    from test_process_manager_cpu import compare_with_golden, run_pm
    from pydem_amd import _ffi
    from pydem_amd.parallel import RcclTransport
    g = load_golden('pm_fractal_2x3_ov1')
    comm = _ffi.Comm(1, 0, _ffi.Comm.unique_id(), 0)

    class T(RcclTransport):
        def __init__(self, pm):
            RcclTransport.__init__(self, pm, comm)

    from pydem_amd import process_manager
    orig = process_manager.EdgeTransport
    process_manager.EdgeTransport = T          # ProcessManager builds its default transport from this name
    try:
        pm, compact, order = run_pm(g, str(tmp_path), n_workers=n_workers)
    finally:
        process_manager.EdgeTransport = orig
    assert isinstance(pm.transport, RcclTransport)
    if n_workers == 1:
        compare_with_golden(pm, compact, order, g, _close)
    else:
        pm2, compact2, _ = run_pm(g, str(tmp_path / 'inproc'), n_workers=n_workers)
        assert (pm.edge_waves, pm.edge_rounds) == (pm2.edge_waves, pm2.edge_rounds)
        for key in compact:
            assert np.array_equal(compact[key], compact2[key], equal_nan=True), key
    assert pm.transport.allreduce_max(3.5) == 3.5
    comm.close()


REF_CASES = [((3, 3), 2), ((4, 5), 2), ((4, 5), 3), ((3, 3), 1), ((4, 3), 1)]    # test_end_to_end.py:86-149, (ny, nx), overlap


@pytest.mark.parametrize('tiles,ov', REF_CASES)
def test_pool_mode_reference_acceptance_cases(tiles, ov, tmp_path):
    """edge_mode='pool' on the device: the reference's own acceptance test (stitched multi-tile UCA == single-tile UCA
    on [1:-1, 1:-1], 6 decimals; NN = 32 like its class) with the rounds of a wave running side by side."""
    import os
    from pydem_amd import DEMProcessor, process_manager, synth
    cone = synth.cone_scaled(32)
    single = DEMProcessor(elev=cone, fill_flats=False, drain_pits_path=False)
    single.dX[:] = 1; single.dY[:] = 1; single.dX2[:] = 1; single.dY2[:] = 1
    single.calc_twi()
    d = str(tmp_path)
    for t, (elev, bounds) in enumerate(synth.split_mosaic(cone, tiles[0], tiles[1], ov)):
        np.savez(os.path.join(d, 'tile_%03d.npz' % t), elev=elev, bounds=bounds)
    process_manager.DEBUG = True
    try:
        pm = process_manager.ProcessManager(in_path=d, elev_conditioned=True, n_workers=8, tiles_in_flight=4)
        pm.process_twi()
        compact = pm.save_non_overlap_data()
    finally:
        process_manager.DEBUG = False
    assert pm.edge_waves < pm.edge_rounds
    np.testing.assert_array_almost_equal(compact['uca'][1:-1, 1:-1], single.uca[1:-1, 1:-1], decimal=6)


@pytest.mark.parametrize('name', ['pm_fractal_2x3_ov1', 'pm_fractal_2x2_ov2', 'pm_nansea_2x2_ov2', 'pm_cone32_4x5_ov3'])
def test_pool_mode_device_matches_oracle_backed_flow(name, tmp_path):
    """Same waves, same masks, UCA / TWI within 1e-9 of the same host logic run with the oracle-backed processor."""
    from oracle_processor import OracleProcessor
    from test_process_manager_cpu import run_pm
    g = load_golden(name)
    pm, compact, _ = run_pm(g, str(tmp_path / 'dev'), n_workers=8, tiles_in_flight=2)
    po, compact_o, _ = run_pm(g, str(tmp_path / 'ora'), n_workers=8, processor_cls=OracleProcessor)
    assert (pm.edge_waves, pm.edge_rounds, pm.edge_tiebreaks) == (po.edge_waves, po.edge_rounds, po.edge_tiebreaks)
    for i in range(pm.n_inputs):
        assert np.array_equal(pm.tile_result(i, 'edge_todo'), po.tile_result(i, 'edge_todo')), i
        assert np.array_equal(pm.tile_result(i, 'edge_done'), po.tile_result(i, 'edge_done')), i
        _close(pm.tile_result(i, 'uca_total'), po.tile_result(i, 'uca_total'), 'tile %d uca' % i)
    for key in ('uca', 'twi'):
        a, b = compact[key], compact_o[key]
        finite = np.isfinite(a) & np.isfinite(b)
        assert np.array_equal(np.isnan(a), np.isnan(b)), key
        assert np.allclose(a[finite], b[finite], rtol=1e-9, atol=1e-12), key


@pytest.mark.parametrize('stop_after,n_workers', [(1, 1), (2, 1), (2, 8), (3, 8)])
def test_resume_from_tile_store_on_device(stop_after, n_workers, tmp_path):
    """`checkpoint=True`: a directory job that stopped after a phase continues from the `.npy` tile store (the reference's
    `success` table, process_manager.py:998-1007 ...) -- resumed tiles get their fields uploaded, the flow graph is rebuilt on
    the device before the first edge round -- and ends with the results of an uninterrupted run."""
    from test_process_manager_grid import write_tiles
    from pydem_amd import process_manager
    g = load_golden('pm_fractal_2x3_ov1')
    src = str(tmp_path / 'tiles')
    write_tiles(g, src, key='elev')
    phases = ['process_elevation', 'process_aspect_slope', 'process_uca', 'process_uca_edges']
    process_manager.DEBUG = True
    try:
        mk = lambda out, **kw: process_manager.ProcessManager(in_path=src, out_path=out, elev_conditioned=True,
                                                              dem_proc_kwargs={'drain_pits': True}, n_workers=n_workers, **kw)
        ref = mk(str(tmp_path / 'plain'))
        ref.process_twi()
        want = ref.save_non_overlap_data()
        first = mk(str(tmp_path / 'store'), checkpoint=True)
        first.compute_grid()
        for name in phases[:stop_after + 1]:
            getattr(first, name)()
        again = mk(str(tmp_path / 'store'), checkpoint=True)
        again.process_twi()
        got = again.save_non_overlap_data()
    finally:
        process_manager.DEBUG = False
    for key in want:
        _close(got[key], want[key], key)


@pytest.mark.parametrize('name,world', [('pm_fractal_2x3_ov1', 2), ('pm_cone32_4x5_ov3', 2),
                                        ('pm_cone32_4x5_ov3', 8),        # the node's real world size: 20 tiles over 8 ranks
                                        ('pm_fractal_2x3_ov1', 8)])      # ... and more ranks than tiles (two ranks own nothing)
def test_two_rank_pool_mode_with_device_board(name, world, tmp_path):
    """N > 1 with the strips on the device: `world` processes (tiles i % world, all on this box's one GPU -- RCCL refuses
    two ranks on one device, so the board's staging buffer is summed over the ranks through the socket group:
    pydem_board_refresh_stage / _unstage), replicated edge board, deterministic waves.  Every rank must see the waves,
    round counts and per-tile results of the single-process pool run.  World size 8 runs every rank-count-dependent
    branch of the N = 8 path (partition, collective board decision, staging offsets, ranks without tiles) except
    ncclAllReduce itself (reference: the worker pool of pydem/process_manager.py:243-255, 1214-1246)."""
    import subprocess
    import sys
    from conftest import ROOT
    from test_process_manager_grid import write_tiles
    g = load_golden(name)
    write_tiles(g, str(tmp_path), key='elev')
    from pydem_amd.rendezvous import spawn_ranks
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    rc, out = spawn_ranks([sys.executable, os.path.join(ROOT, 'tests', '_dist_pm_worker.py'), name, str(tmp_path), 'pool', 'device'], world,
                          env=env, master_port=29600 + (os.getpid() % 300), capture=True, timeout=900)
    assert rc == 0, out[-3000:]
    assert out.count(' ok: ') == world, out[-3000:]


@pytest.mark.parametrize('name,world', [('pm_fractal_2x2_ov2', 2), ('pm_nansea_2x2_ov2', 2), ('pm_fractal_2x3_ov1', 8), ('pm_fractal_2x2_ov2', 8)])
def test_multi_rank_queued_waves_over_the_socket_group(name, world, tmp_path):
    """The QUEUED waves of the fix-up (pydem_board_run_waves_ex) with more than one rank on this box's one GPU: every rank
    runs the wave-selection kernel from its replica of the board, the byte staging buffer is summed over the socket group
    where the RCCL path calls ncclAllReduce(ncclUint8), and before every sum the ranks check that they selected the SAME
    wave.  Mosaics of at most 2 * world tiles (the queue's condition); the worker holds waves, rounds, tie-breaks and
    every owned tile against the single-process pool run and asserts that batches were queued on its rank (reference: the
    manager loop pydem/process_manager.py:1214-1246)."""
    import sys
    from conftest import ROOT
    from test_process_manager_grid import write_tiles
    g = load_golden(name)
    write_tiles(g, str(tmp_path), key='elev')
    from pydem_amd.rendezvous import spawn_ranks
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1', PYDEM_EXPECT_QUEUED='1')
    rc, out = spawn_ranks([sys.executable, os.path.join(ROOT, 'tests', '_dist_pm_worker.py'), name, str(tmp_path), 'pool', 'device'], world,
                          env=env, master_port=29900 + (os.getpid() % 90), capture=True, timeout=900)
    assert rc == 0, out[-3000:]
    assert out.count(' ok: ') == world, out[-3000:]


def test_ranks_that_disagree_on_a_wave_stop_with_a_message(tmp_path):
    """pydem_board_run_waves_ex compares the selected wave over the ranks before it sums the staging buffer: an exchange whose
    maximum reports a wave this rank did not select makes the batch return an error that names the wave, instead of queueing
    on (with RCCL such ranks would sit in ncclAllReduce; there the watchdog PYDEM_EDGE_TIMEOUT ends the wait)."""
    import warnings
    from test_process_manager_grid import write_tiles
    from pydem_amd import _ffi, process_manager
    g = load_golden('pm_fractal_2x2_ov2')
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}

    class Liar(process_manager.EdgeTransport):
        # a transport of one rank whose "maximum over the ranks" claims that somebody selected another wave
        def sum_inplace(self, arr):
            pass

        def sum_bytes_inplace(self, arr):
            pass

        def max_inplace(self, arr):
            arr[0] += 1.0
    write_tiles(g, str(tmp_path), key='elev')
    process_manager.DEBUG = True
    try:
        pm = process_manager.ProcessManager(in_path=str(tmp_path), dem_proc_kwargs=dkw, elev_conditioned=True, n_workers=8, transport=False)
        pm.transport = Liar(pm)
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            with pytest.raises(_ffi.HipError, match='disagree on wave'):
                pm.process_twi()
    finally:
        process_manager.DEBUG = False


@pytest.mark.parametrize('world', [2, 8])
@pytest.mark.parametrize('name', ['pm_fractal_2x3_ov1', 'pm_cone32_4x5_ov3'])
def test_two_rank_pool_mode_over_rccl(name, world, tmp_path):
    """The same with one GPU per rank and a real communicator of world size 2: `pydem_board_refresh` replicates the
    staging buffer with ncclAllReduce (csrc/comm.hip), the RCCL id travels over pydem_amd.rendezvous.  Needs two GPUs
    (the driver's single-GPU test box skips it; reference: the strips of pydem/process_manager.py:243-255)."""
    import sys
    from conftest import ROOT
    from pydem_amd import _ffi
    from pydem_amd.rendezvous import spawn_ranks
    from test_process_manager_grid import write_tiles
    if _ffi.device_count() < world:
        pytest.skip("needs %d GPUs (RCCL refuses two ranks on one device)" % world)
    g = load_golden(name)
    write_tiles(g, str(tmp_path), key='elev')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    rc, out = spawn_ranks([sys.executable, os.path.join(ROOT, 'tests', '_dist_pm_worker.py'), name, str(tmp_path), 'pool', 'rccl'], world,
                          env=env, master_port=29300 + (os.getpid() % 300), capture=True, timeout=900)
    assert rc == 0, out[-3000:]
    assert out.count(' ok: ') == world, out[-3000:]


@pytest.mark.parametrize('name', ['pm_fractal_2x3_ov1', 'pm_fractal_2x2_ov2', 'pm_nansea_2x2_ov2'])
@pytest.mark.parametrize('k', [1, 3, 16])
def test_queued_waves_equal_host_driven_waves(name, k, tmp_path, monkeypatch):
    """The waves chosen by a kernel and queued k at a time (pydem_board_run_waves) against the host-driven wave loop
    (PYDEM_EDGE_QUEUE=0): the same members wave by wave, the same tie-breaks, masks and areas identical bit for bit (the
    rounds are the same kernels in the same order), and fewer looks from the host."""
    from test_process_manager_cpu import run_pm
    g = load_golden(name)
    monkeypatch.setenv('PYDEM_EDGE_QUEUE', '0')
    p0, c0, _ = run_pm(g, str(tmp_path / 'host'), n_workers=8, tiles_in_flight=2)
    monkeypatch.setenv('PYDEM_EDGE_QUEUE', str(k))
    pk, ck, _ = run_pm(g, str(tmp_path / 'queued'), n_workers=8, tiles_in_flight=2)
    assert (pk.edge_waves, pk.edge_rounds, pk.edge_tiebreaks) == (p0.edge_waves, p0.edge_rounds, p0.edge_tiebreaks)
    assert sorted((w, a) for w, a, _ in pk.edge_round_log) == sorted((w, a) for w, a, _ in p0.edge_round_log)    # (rounds of a wave finish in any order)
    assert p0.edge_host_looks == p0.edge_waves + 1 and p0.edge_queued_batches == 0
    # (until round 6 the queue was silently off under the default keep_first_pass_uca=True and this test compared the host loop
    # with itself: the batches are counted now)
    assert pk.edge_queued_batches > 0
    if k > 1:
        assert pk.edge_host_looks <= p0.edge_host_looks    # (small mosaics: a tile's first round still goes to the host; batches
                                                           #  of ONE wave pay a look per wave plus one per batch that stops for the host)
    for i in range(pk.n_inputs):
        for key in ('edge_todo', 'edge_done'):
            assert np.array_equal(pk.tile_result(i, key), p0.tile_result(i, key)), (i, key)
        assert np.array_equal(pk.tile_result(i, 'uca_total'), p0.tile_result(i, 'uca_total'), equal_nan=True), i
    for key in ck:
        assert np.array_equal(ck[key], c0[key], equal_nan=True), key


def test_queued_waves_respect_the_wave_limit(tmp_path, monkeypatch):
    """max_edge_rounds cuts a batch of queued waves where it cuts the host-driven loop."""
    from test_process_manager_grid import write_tiles
    from pydem_amd import process_manager
    g = load_golden('pm_fractal_2x3_ov1')
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}
    out = []
    for q in ('0', '16'):
        monkeypatch.setenv('PYDEM_EDGE_QUEUE', q)
        d = str(tmp_path / q)
        write_tiles(g, d, key='elev')
        process_manager.DEBUG = True
        try:
            pm = process_manager.ProcessManager(in_path=d, dem_proc_kwargs=dkw, elev_conditioned=True, n_workers=8)
            pm.max_edge_rounds = 3
            pm.process_twi()
        finally:
            process_manager.DEBUG = False
        out.append(pm)
    assert out[0].edge_waves == out[1].edge_waves == 3
    assert out[0].edge_rounds == out[1].edge_rounds
    assert out[0].edge_queued_batches == 0
    for i in range(out[0].n_inputs):
        assert np.array_equal(out[0].tile_result(i, 'uca_total'), out[1].tile_result(i, 'uca_total'), equal_nan=True), i
        assert np.array_equal(out[0].tile_result(i, 'edge_done'), out[1].tile_result(i, 'edge_done')), i


@pytest.mark.parametrize('name', ['pm_fractal_2x3_ov1', 'pm_nansea_2x2_ov2'])
def test_device_operator_build_matches_host_build(name, tmp_path, monkeypatch):
    """The condensed operator of a tile's fix-up (csrc/uca_cond.inl) built on the device (csrc/uca_cbuild.inl, the default since
    round 6) against the host build of rounds 4-5: PYDEM_COND_BUILD=check builds both for every tile and compares node by node
    (cells, counts, edges and in-slots exactly, weights to 1e-12 -- the host orders the pit edges of one pit by record id, the
    device by drain cell) and fails the round on a difference; PYDEM_COND_BUILD=host must give the same waves and masks as the
    default (reference: the round pydem/dem_processing.py:778-862 behind pydem/process_manager.py:224-284)."""
    from test_process_manager_cpu import run_pm
    g = load_golden(name)
    runs = {}
    for how in ('device', 'check', 'host', 'fallback'):
        # ('fallback': the device build gives up after its sweep -- as it would on a merge larger than its staging area or a pool
        # region that overflows -- and the host build takes the tile over)
        monkeypatch.setenv('PYDEM_COND_BUILD', 'device' if how == 'fallback' else how)
        monkeypatch.setenv('PYDEM_CB_FORCE_FALLBACK', '1' if how == 'fallback' else '0')
        runs[how] = run_pm(g, str(tmp_path / how), n_workers=8)[0]
    d = runs['device']
    assert d.edge_queued_batches > 0
    for how in ('check', 'host', 'fallback'):
        p = runs[how]
        assert (p.edge_waves, p.edge_rounds, p.edge_tiebreaks) == (d.edge_waves, d.edge_rounds, d.edge_tiebreaks), how
        for i in range(d.n_inputs):
            for key in ('edge_todo', 'edge_done'):
                assert np.array_equal(p.tile_result(i, key), d.tile_result(i, key)), (how, i, key)
            a, b = p.tile_result(i, 'uca_total'), d.tile_result(i, 'uca_total')
            assert np.array_equal(np.isnan(a), np.isnan(b)), (how, i)
            assert np.allclose(a, b, rtol=1e-12, atol=0, equal_nan=True), (how, i)


def test_bench_with_two_ranks_on_this_box(tmp_path):
    """bench.py --gpus 2 (it starts its own ranks): without two GPUs it must refuse at once with a message, not hang in a
    communicator; with PYDEM_BENCH_SHARED_GPU=1 the two ranks share the GPU, the strips travel over sockets, every rank chooses the
    waves on its own replica of the board (queued batches on both), the ranks compare schedule and board after the warm-up, and the
    line carries per-rank numbers."""
    import json
    import subprocess
    import sys
    from conftest import ROOT
    from pydem_amd import _ffi
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', MASTER_PORT=str(29700 + os.getpid() % 200))
    env.pop('PYDEM_BENCH_SHARED_GPU', None)
    cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--size', '1024', '--steps', '1', '--warmup', '1', '--cpu-sample', '0', '--roof-iters', '0']
    if _ffi.device_count() < 2:
        r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode != 0 and 'needs 2 GPUs' in (r.stderr + r.stdout), (r.returncode, r.stderr[-500:])
    r = subprocess.run(cmd, env=dict(env, PYDEM_BENCH_SHARED_GPU='1', MASTER_PORT=str(29400 + os.getpid() % 200)), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['rccl_ranks'] is None and line['config']['edge_exchange'] == 'socket-host-fallback'
    ranks = line['per_rank']
    assert [p['rank'] for p in ranks] == [0, 1]
    assert ranks[0]['edge_waves'] == ranks[1]['edge_waves'] > 0
    assert all(p['edge_queued_batches'] > 0 for p in ranks)
