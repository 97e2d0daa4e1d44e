"""Raising stand-in for `rasterio` (not installed). TEST INFRASTRUCTURE ONLY."""


def open(*a, **k):
    raise RuntimeError("rasterio is not available in this image (array-only oracle harness)")
