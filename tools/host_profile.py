#!/usr/bin/env python
"""GPU box: where the host time of a bench step goes (cProfile over a few steps of the single-tile flow)."""
import cProfile
import os
import pstats
import sys
import time
import warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
warnings.simplefilter('ignore')
import bench
from pydem_amd import process_manager
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
pm = process_manager.ProcessManager(elev_source_files=bench.tile_specs(1, n, n), elev_conditioned=True, dem_proc_kwargs={'drain_pits': True},
                                    devices=[0], keep_first_pass_uca=False, n_workers=1, edge_mode='reference')
pm.compute_grid(); pm.process_elevation()


def step():
    pm.process_aspect_slope(); pm.process_uca()
    pm.tiles[0]._tile.synchronize()
    pm.process_uca_edges()
    pm.tiles[0].find_flats(); pm.tiles[0].run_twi()


step(); step()
t0 = time.perf_counter()
for _ in range(3):
    step()
print('ms per step %.2f' % ((time.perf_counter() - t0) / 3 * 1e3), pm.tiles[0]._tile.timings())
pr = cProfile.Profile()
pr.enable()
for _ in range(3):
    step()
pr.disable()
pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
