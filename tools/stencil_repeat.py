import sys, os, warnings
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
warnings.simplefilter('ignore')
import bench
from pydem_amd import process_manager
n = 16384
pm = process_manager.ProcessManager(elev_source_files=bench.tile_specs(1, n, n), elev_conditioned=True,
                                    dem_proc_kwargs={'drain_pits': True}, devices=[0], keep_first_pass_uca=False)
pm.compute_grid(); pm.process_elevation()
t = pm.tiles[0]
for rep in range(2):
    pm.process_aspect_slope(); a = t._tile.timings()['stencil_kernel_ms']
    t._tile.slopes_directions(); b = t._tile.timings()['stencil_kernel_ms']
    t._tile.slopes_directions(); c = t._tile.timings()['stencil_kernel_ms']
    pm.process_uca(); t.find_flats(); t.run_twi()
    print('after a full step: %.3f ms; immediately again: %.3f, %.3f ms' % (a, b, c))
