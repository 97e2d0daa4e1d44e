"""N > 1 path on CPU: two processes (gloo), tiles sharded i % 2, edge strips through
pydem_amd.parallel.DistTransport -- the same protocol the RCCL transport runs on the GPU box.
The per-tile arithmetic is the oracle-backed processor (tests/oracle_processor.py)."""
import os
import subprocess
import sys

import pytest

from conftest import ROOT, load_golden
from test_process_manager_grid import write_tiles


@pytest.mark.parametrize('mode', ['reference', 'pool'])
@pytest.mark.parametrize('name', ['pm_cone32_3x3_ov1', 'pm_fractal_2x3_ov1'])
def test_two_rank_directory_flow(name, mode, tmp_path):
    g = load_golden(name)
    write_tiles(g, str(tmp_path), key='elev')
    env = dict(os.environ, MASTER_ADDR='127.0.0.1', OMP_NUM_THREADS='1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node=2', '--master-addr', '127.0.0.1',
           '--master-port', str(29500 + (os.getpid() % 500)), os.path.join(ROOT, 'tests', '_dist_pm_worker.py'), name, str(tmp_path), mode]
    res = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=300)
    out = res.stdout.decode()
    assert res.returncode == 0, out[-3000:]
    assert out.count(' ok: ') == 2, out[-3000:]
