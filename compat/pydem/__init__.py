"""Import-name compatibility: put `<repo>/compat` on PYTHONPATH and code written against the reference
(`from pydem.dem_processing import DEMProcessor`, `from pydem.process_manager import ProcessManager`,
`from pydem.cyfuncs import cyutils`) runs on the MI355X path of pydem_amd unchanged.  Nothing is
implemented here: the modules re-export pydem_amd (reference layout: pydem/__init__.py,
pydem/dem_processing.py, pydem/process_manager.py, pydem/cyfuncs/cyutils.pyx)."""
import os
import sys

_root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if _root not in sys.path:
    sys.path.insert(0, _root)

from pydem_amd import DEMProcessor  # noqa: E402,F401  (pydem/__init__.py exports DEMProcessor and the process_manager module)
from pydem_amd.process_manager import ProcessManager  # noqa: E402,F401
from . import process_manager  # noqa: E402,F401
