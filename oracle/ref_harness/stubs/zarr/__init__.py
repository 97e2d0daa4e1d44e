"""Raising stand-in for `zarr` (not installed). TEST INFRASTRUCTURE ONLY."""


def open(*a, **k):
    raise RuntimeError("zarr is not available in this image (array-only oracle harness)")
