"""pydem.process_manager -> pydem_amd.process_manager."""
from pydem_amd.process_manager import *  # noqa: F401,F403
from pydem_amd.process_manager import ProcessManager  # noqa: F401
