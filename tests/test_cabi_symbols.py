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


def test_queued_wave_state_layout_is_the_same_on_both_sides():
    """The schedule state of pydem_board_run_waves travels as a flat uint64 array: the word indices of csrc/comm.hip (SCH_*),
    of the ctypes wrapper (_ffi.Board.SCH_*) and the word count quoted in include/pydem_hip.h must agree."""
    import os
    import re
    from pydem_amd import _ffi
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = open(os.path.join(root, 'pydem_amd', 'csrc', 'comm.hip')).read()
    enum = re.search(r'enum \{ (SCH_OK = 0.*?) \};', src, re.S).group(1)
    c_side = {k: int(v) for k, v in re.findall(r'(SCH_[A-Z]+) = (\d+)', enum)}
    py_side = {k: getattr(_ffi.Board, k) for k in dir(_ffi.Board) if k.startswith('SCH_')}
    assert py_side and all(c_side[k] == v for k, v in py_side.items()), (c_side, py_side)
    header = open(os.path.join(root, 'include', 'pydem_hip.h')).read()
    assert '`state`: %d 64-bit words' % c_side['SCH_WORDS'] in header
    # per-tile ranges hold 64 tiles and do not overlap
    starts = sorted(v for k, v in c_side.items() if k in ('SCH_ND', 'SCH_PD', 'SCH_HASH', 'SCH_HAS', 'SCH_READERS', 'SCH_NBRS', 'SCH_LOG', 'SCH_ROUND', 'SCH_TB'))
    assert all(b - a >= 64 for a, b in zip(starts, starts[1:]))
