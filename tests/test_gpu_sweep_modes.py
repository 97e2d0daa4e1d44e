"""The alternative sweep schedules (the LDS-resident first pass, PYDEM_SWEEP_FIRST=lds; resident / generic visits; the two-level
solve of the tail at several switch points, PYDEM_SWEEP_SYM) must give the same answers as the default tile-pass schedule.  The mode is read once per process, hence the subprocess.  The frontier-queue
schedule (PYDEM_SWEEP_MODE=queue) only exists in a diagnostic build of the library (PYDEM_HIPCC_FLAGS=-DPYDEM_SWEEP_QUEUE: it
refuses some valid inputs, so a product build ignores the variable); its cases run with PYDEM_TEST_SWEEP_QUEUE=1."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

SCRIPT = r'''
import sys, warnings
import numpy as np
sys.path.insert(0, %(root)r)
warnings.simplefilter('ignore')
from oracle import oracle as O
from pydem_amd import DEMProcessor, synth
for shape, seed in (((700, 520), 41), ((1024, 1024), 42)):
    z = synth.fractal(shape[0], shape[1], seed=seed, top_shift=7, n_octaves=7)
    o = O.OracleDEM(z, dX=30.0, dY=30.0, drain_pits=True); o.calc_uca()
    dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=False, drain_pits_path=False, drain_pits=True)
    dp.calc_slopes_directions(); dp.calc_uca()
    assert np.array_equal(np.isnan(dp.uca), np.isnan(o.uca))
    assert np.allclose(dp.uca, o.uca, rtol=1e-9, atol=0, equal_nan=True)
    assert np.array_equal(dp.edge_todo, o.edge_todo) and np.array_equal(dp.edge_done, o.edge_done)
print("MODES-OK", dp.timings['sweep_rounds'], dp.timings['sweep_kernel_launches'])
'''


QUEUE_BUILD = os.environ.get('PYDEM_TEST_SWEEP_QUEUE') == '1'
needs_queue_build = pytest.mark.skipif(not QUEUE_BUILD, reason="the queue schedule needs a library built with -DPYDEM_SWEEP_QUEUE (PYDEM_TEST_SWEEP_QUEUE=1)")


@pytest.mark.parametrize('env', [pytest.param({'PYDEM_SWEEP_MODE': 'queue'}, marks=needs_queue_build),
                                 pytest.param({'PYDEM_SWEEP_MODE': 'queue', 'PYDEM_SWEEP_TILE_SWITCH': '0'}, marks=needs_queue_build),
                                 pytest.param({'PYDEM_SWEEP_MODE': 'queue', 'PYDEM_TILE_PASSES': '3'}, marks=needs_queue_build),
                                 {'PYDEM_SWEEP_FIRST': 'lds'},           # pass 1 by the LDS-resident kernel (csrc/uca.hip K5a)
                                 {'PYDEM_SWEEP_DENSE': '3'},             # three dense level kernels ahead of the tile passes (K5d)
                                 {'PYDEM_SWEEP_SYM': '0'},               # tile passes only: no two-level solve for the tail (K5f, csrc/uca_sym.inl)
                                 {'PYDEM_SWEEP_SYM': '100000000'},       # the symbolic pass right after the two full passes
                                 {'PYDEM_SWEEP_SYM': '40'},              # ... late: when at most 40 tiles are listed
                                 {'PYDEM_SWEEP_SYM': '100000000', 'PYDEM_SWEEP_RESIDENT': '0'},
                                 {'PYDEM_SWEEP_RESIDENT': '0'},          # generic visits in every listed pass
                                 {'PYDEM_SWEEP_RESIDENT': '100000000'}]) # resident visits (K5e) from pass 3 on, tiles with > 256 open cells generic
def test_queue_schedule_matches_oracle(env):
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-c', SCRIPT % {'root': root}], env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and 'MODES-OK' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@needs_queue_build
def test_circular_drainage_replay_in_queue_schedule():
    """The re-seed replay over the cells on / below a drainage loop (csrc/uca.hip K5c) runs after either schedule: the
    hand-made loop fields of tests/test_gpu_parity.py, once more with PYDEM_SWEEP_MODE=queue."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PYDEM_SWEEP_MODE='queue')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_parity.py'), '-q', '-x', '-k', 'circular_drainage'],
                       env=e, capture_output=True, text=True, timeout=600, cwd=root)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.parametrize('env', [{'PYDEM_FLAT_COOP': '0'},                       # fill_flats sweeps: a launch per sweep / one workgroup, as in round 2
                                 {'PYDEM_FLAT_COOP': '1000000', 'PYDEM_FLAT_COOP_WG': '7'},     # resident workgroups from the second look on
                                 {'PYDEM_PATHS_MID': '0'},                       # pit drain paths without the medium window
                                 {'PYDEM_PATHS_WINDOW': '64', 'PYDEM_PATHS_BIG': '2'},          # tiny speculation windows: many rounds, capped large simulations
                                 {'PYDEM_PATHS_BIG': '8', 'PYDEM_PATHS_LARGE': '1'},            # one large-window simulation per round: the others wait
                                 {'PYDEM_PATHS_KEEP': '0'},                      # every waiting pit simulated again every round (rounds 1-4)
                                 {'PYDEM_PATHS_WINDOW': '131072', 'PYDEM_PATHS_BIG': '3'}])   # kept simulations with a medium-window pool of three blocks
def test_conditioning_schedules_match_the_host_twin(env):
    """The schedule switches of the device conditioning (csrc/cond_device.hip, csrc/cond_paths.hip) change how the work is
    cut, never the result: the conditioning tests once more per setting (read once per process, hence the subprocess)."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ); e.update(env)
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_conditioning.py'), '-q', '-x'],
                       env=e, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]


def test_circular_drainage_replay_on_the_host():
    """Above 4 M unfinished cells (a drainage loop at the head of a long river) the re-seed loop of the reference
    (pydem/dem_processing.py:951-964, cyutils.pyx:119-187) is replayed by the HOST instead of one GPU thread -- the tile is
    finished like the reference finishes it, not refused.  PYDEM_RESEED_HOST_ABOVE=0 sends the hand-made loop fields of
    tests/test_gpu_parity.py and the soak's mosaic with circular drainage down that path."""
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    e = dict(os.environ, PYDEM_RESEED_HOST_ABOVE='0')
    r = subprocess.run([sys.executable, '-m', 'pytest', os.path.join(root, 'tests', 'test_gpu_parity.py'), os.path.join(root, 'tests', 'test_gpu_soak.py'),
                        '-q', '-x', '-s', '-k', 'circular'], env=e, capture_output=True, text=True, timeout=900, cwd=root)
    assert r.returncode == 0 and ' passed' in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
    assert 'the re-seed loop runs on the host' in r.stdout + r.stderr, "the host replay did not run"
