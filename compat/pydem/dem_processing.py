"""pydem.dem_processing -> pydem_amd.dem_processing (same class, options and defaults)."""
from pydem_amd.dem_processing import *  # noqa: F401,F403
from pydem_amd.dem_processing import DEMProcessor  # noqa: F401
