def distance(*a, **k):
    raise RuntimeError("geopy is not available in this image (array-only oracle harness)")
