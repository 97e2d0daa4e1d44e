from pydem_amd.cyfuncs import cyutils  # noqa: F401
