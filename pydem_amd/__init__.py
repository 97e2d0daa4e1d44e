"""pydem_amd -- MI355X-native implementation of pyDEM's per-tile terrain hot path
(slope/aspect, D-infinity flow graph, upstream contributing area, TWI) behind the reference's
DEMProcessor / ProcessManager Python API.  HIP kernels via a ctypes C-ABI; no torch."""
__version__ = "0.1.0"

from .dem_processing import DEMProcessor  # noqa: F401
