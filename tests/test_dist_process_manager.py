"""N > 1 path on CPU: two processes, tiles sharded i % 2, edge strips through pydem_amd.parallel.DistTransport (socket group of
pydem_amd.rendezvous: no framework) -- the same protocol the RCCL transport runs on the GPU box.
The per-tile arithmetic is the oracle-backed processor (tests/oracle_processor.py)."""
import os
import sys

import pytest

from conftest import ROOT, load_golden
from test_process_manager_grid import write_tiles


@pytest.mark.parametrize('mode', ['reference', 'pool'])
@pytest.mark.parametrize('name', ['pm_cone32_3x3_ov1', 'pm_fractal_2x3_ov1'])
def test_two_rank_directory_flow(name, mode, tmp_path):
    g = load_golden(name)
    write_tiles(g, str(tmp_path), key='elev')
    from pydem_amd.rendezvous import spawn_ranks
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    rc, out = spawn_ranks([sys.executable, os.path.join(ROOT, 'tests', '_dist_pm_worker.py'), name, str(tmp_path), mode], 2, env=env,
                          master_port=29500 + (os.getpid() % 500), capture=True, timeout=300)
    assert rc == 0, out[-3000:]
    assert out.count(' ok: ') == 2, out[-3000:]


def test_more_ranks_than_tiles(tmp_path):
    """world_size > n_tiles: tile i -> rank i % world leaves the last rank without a tile.  The choice between the host pool
    path and the device edge board is collective (ProcessManager._device_board_usable), so the rank without tiles takes
    the same branch as the others and the job ends instead of hanging in mismatched collectives."""
    from pydem_amd.rendezvous import spawn_ranks
    name, world = 'pm_fractal_2x2_ov2', 5            # four tiles, five ranks
    g = load_golden(name)
    write_tiles(g, str(tmp_path), key='elev')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    rc, out = spawn_ranks([sys.executable, os.path.join(ROOT, 'tests', '_dist_pm_worker.py'), name, str(tmp_path), 'pool'], world, env=env,
                          master_port=29000 + (os.getpid() % 400), capture=True, timeout=300)
    assert rc == 0, out[-3000:]
    assert out.count(' ok: ') == world, out[-3000:]
