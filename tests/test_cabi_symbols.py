"""CPU tier: libpydem_hip.so loads without a GPU and exports every function include/pydem_hip.h declares;
the ctypes binding table covers the same set; calling into the library without a device fails loudly
(no CPU fallback)."""
import ctypes
import os
import re

import pytest

from conftest import ROOT


def _declared():
    text = open(os.path.join(ROOT, 'include', 'pydem_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(pydem_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from pydem_amd import _ffi, build
    build.build(verbose=False)
    lib = ctypes.CDLL(_ffi.LIB_PATH)
    names = _declared()
    assert len(names) >= 30
    for nm in names:
        assert hasattr(lib, nm), "libpydem_hip.so does not export %s" % nm
    assert sorted(_ffi.SYMBOLS) == names, "ctypes table and header disagree: %s" % (set(_ffi.SYMBOLS) ^ set(names))


def test_no_cpu_fallback_without_a_device():
    from pydem_amd import _ffi
    try:
        n = _ffi.device_count()
    except _ffi.HipError:
        n = 0
    if n > 0:
        pytest.skip("a GPU is visible")
    import numpy as np
    from pydem_amd import DEMProcessor
    dp = DEMProcessor(elev=np.arange(25.0).reshape(5, 5), fill_flats=False, drain_pits_path=False)
    with pytest.raises(_ffi.HipError):
        dp.calc_slopes_directions()
