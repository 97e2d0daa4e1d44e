import sys, time, warnings, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
import bench
from pydem_amd import process_manager
n = int(sys.argv[1]); nt = int(sys.argv[2])
pm = process_manager.ProcessManager(elev_source_files=bench.tile_specs(nt, n, n), elev_conditioned=True,
                                    dem_proc_kwargs={'drain_pits': True}, devices=[0], keep_first_pass_uca=False)
pm.compute_grid(); pm.process_elevation()
pm.process_aspect_slope(); pm.process_uca(); pm.process_uca_edges()
pm.process_aspect_slope(); pm.process_uca()
import cProfile, pstats
pr = cProfile.Profile(); pr.enable(); t0 = time.perf_counter(); pm.process_uca_edges(); dt = time.perf_counter() - t0; pr.disable()
print('edge fix-up %.1f ms, %d rounds (%d skipped)' % (dt * 1e3, pm.edge_rounds, pm.edge_rounds_skipped))
pstats.Stats(pr).sort_stats('tottime').print_stats(14)
