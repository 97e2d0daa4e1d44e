"""pydem.cyfuncs.cyutils -> pydem_amd.cyfuncs.cyutils (drain_area / drain_connections on the GPU)."""
from pydem_amd.cyfuncs.cyutils import drain_area, drain_connections  # noqa: F401
