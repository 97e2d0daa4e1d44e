import sys, time, warnings
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
import bench
from pydem_amd import process_manager
n = int(sys.argv[1]); nt = int(sys.argv[2])
pm = process_manager.ProcessManager(elev_source_files=bench.tile_specs(nt, n, n), elev_conditioned=True,
                                    dem_proc_kwargs={'drain_pits': os.environ.get('PM_DRAIN', '1') == '1'}, devices=[0], keep_first_pass_uca=False,
                                    tiles_in_flight=(None if os.environ.get('PM_IN_FLIGHT') == 'auto' else int(os.environ.get('PM_IN_FLIGHT', '1'))),   # auto = the manager's default: builds / flushes of the fix-up side by side
                                    n_workers=int(os.environ.get('PM_WORKERS', '1')), edge_mode=os.environ.get('PM_EDGE_MODE') or None)
if os.environ.get('PM_RCCL'):      # the RCCL strip transport with a single rank: its per-round overhead against the in-process one
    from pydem_amd import _ffi
    from pydem_amd.parallel import RcclTransport
    pm.transport = RcclTransport(pm, _ffi.Comm(1, 0, _ffi.Comm.unique_id(), 0))
pm.compute_grid(); pm.process_elevation()
if os.environ.get('PICKS'):
    _orig = pm._edge_round
    picks = []
    def _wrap(i):
        picks.append(i); return _orig(i)
    pm._edge_round = _wrap
for rep in range(2):
    t0 = time.perf_counter(); pm.process_aspect_slope(); pm.process_uca()
    for t in pm.tiles: t._tile.synchronize()
    t1 = time.perf_counter(); pm.process_uca_edges(); t2 = time.perf_counter()
    for t in pm.tiles: t.find_flats(); t.run_twi()
    t3 = time.perf_counter()
    if os.environ.get('PICKS'): print('picks', picks); del picks[:]
    waves = {}
    for w, i, ms in pm.edge_round_log: waves.setdefault(w, []).append(ms)
    queued = getattr(pm, 'edge_host_looks', len(waves) + 1) < len(waves) + 1
    print('   rounds: sum %.1f ms; sum over waves of the slowest round %.1f ms (%s); per wave (tiles, max ms): %s'
          % (sum(ms for _, _, ms in pm.edge_round_log), sum(max(v) for v in waves.values()),
             'HOST-DRIVEN waves only: the queued waves have no per-round times, see the "queued" part of PYDEM_EDGE_PROFILE=1' if queued
             else 'critical path with one tile per GPU', ' '.join('%d:%.1f' % (len(v), max(v)) for v in waves.values())))
    print('   host looks inside the wave loop: %s (queued waves as graphs: %s)' % (getattr(pm, 'edge_host_looks', 'n/a'), getattr(pm, 'edge_wave_graphs', 'n/a')))
    print('n=%d tiles=%d: tiles %.1f ms, edge fix-up %.1f ms (%d rounds in %d waves), twi %.1f ms' % (n, nt, (t1-t0)*1e3, (t2-t1)*1e3, pm.edge_rounds, pm.edge_waves, (t3-t2)*1e3))
if len(sys.argv) > 3:
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable(); pm.process_aspect_slope(); pm.process_uca(); pm.process_uca_edges(); pr.disable()
    print('skipped rounds', pm.edge_rounds_skipped, 'of', pm.edge_rounds)
    pstats.Stats(pr).sort_stats('cumulative').print_stats(18)
