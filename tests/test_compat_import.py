"""The import-name shim: code written against the reference's package name runs on pydem_amd when
<repo>/compat is on PYTHONPATH (reference layout: pydem/dem_processing.py, pydem/process_manager.py,
pydem/cyfuncs/cyutils.pyx).  Import only -- no device work."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_reference_import_names_resolve_to_pydem_amd():
    code = (
        "import pydem, pydem_amd\n"
        "from pydem.dem_processing import DEMProcessor\n"
        "from pydem.process_manager import ProcessManager\n"
        "from pydem.cyfuncs import cyutils\n"
        "import pydem_amd.process_manager as pm, pydem_amd.cyfuncs.cyutils as cy\n"
        "assert DEMProcessor is pydem_amd.DEMProcessor and pydem.DEMProcessor is DEMProcessor\n"
        "assert ProcessManager is pm.ProcessManager\n"
        "assert cyutils.drain_area is cy.drain_area and cyutils.drain_connections is cy.drain_connections\n"
        "print('ok')\n")
    env = dict(os.environ, PYTHONPATH=os.path.join(ROOT, 'compat'))
    out = subprocess.run([sys.executable, '-c', code], env=env, cwd='/tmp', capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and out.stdout.strip() == 'ok', out.stderr


def test_cyutils_shim_checks_its_arguments_like_the_typed_cython_signature():
    """The library reads these arrays through raw pointers: wrong dtypes / sizes must be refused in Python, before
    the call (cyutils.pyx:78-83 declares float64 / uint8 / int32 buffers).  No GPU needed: the checks come first."""
    import numpy as np
    import pytest
    from pydem_amd.cyfuncs import cyutils
    n, m = 4, 5
    N = n * m
    ok = dict(area=np.zeros(N), done=np.zeros(N, bool), ids=np.zeros(N, bool), col_indptr=np.zeros(N + 1, np.int32),
              col_indices=np.zeros(0, np.int32), col_data=np.zeros(0), row_indptr=np.zeros(N + 1, np.int32),
              row_indices=np.zeros(0, np.int32), n_rows=n, n_cols=m)
    bad = [dict(area=np.zeros(N, np.float32)), dict(area=np.zeros(N + 1)), dict(done=np.zeros(N, np.int32)),
           dict(ids=np.zeros(N - 1, bool)), dict(col_indptr=np.zeros(N, np.int32)), dict(row_indptr=np.zeros(N + 2, np.int32)),
           dict(edge_todo=np.zeros(N, bool)), dict(edge_todo=np.zeros((N, 2))[:, 0]), dict(edge_todo_no_mask=np.zeros(N - 1)),
           dict(col_indptr=np.r_[np.zeros(N, np.int32), 3].astype(np.int32))]
    for b in bad:
        with pytest.raises((TypeError, ValueError)):
            cyutils.drain_area(**dict(ok, **b))
    with pytest.raises((TypeError, ValueError)):
        cyutils.drain_connections(np.zeros(N, bool), np.zeros(N, bool), np.zeros(N, np.int32), np.zeros(0, np.int32))
    with pytest.raises((TypeError, ValueError)):
        cyutils.drain_connections(np.zeros(N, np.float64), np.zeros(N, bool), np.zeros(N + 1, np.int32), np.zeros(0, np.int32))
