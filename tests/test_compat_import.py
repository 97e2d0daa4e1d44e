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
