"""Elevation conditioning, host side: quantisation-pit filling, flat interpolation and pit drain paths.

These are the three steps the reference runs before the slope stencil when `fill_flats` /
`drain_pits_path` are set (creare-com/pydem v1.2.1, pydem/dem_processing.py: calc_fill_pit_artifacts
:396-426, calc_fill_flats :551-579 with _fill_flat :308-394, calc_pit_drain_paths :428-548, helpers
pydem/utils.py:270-468).  The product path runs them ON THE DEVICE (csrc/cond_device.hip, csrc/cond_paths.hip,
through DEMProcessor.calc_fill_flats / calc_pit_drain_paths -- tiles with no-data cells included since round 4: the
device replays scipy's ring filter where a NaN is near); this module is what remains on the host: masked arrays,
surfaces of a dtype the device cannot hold (int64), a tile whose no-data cells alternate along a line for longer than
the bounded replay looks back, and the rare tile on which the order-preserving parallel schedule of the pit paths
gives up.  It is also the twin the device path is tested against (its masks come from scipy itself).  The vectorised prologues stay in numpy / scipy.ndimage (3x3 filters,
connected-component labels, the argsort whose tie order is part of the result); the per-region / per-pit loops run in
the native library (csrc/cond_host.cpp, host code behind the same C-ABI).  Results are pinned bit for bit by
tests/golden/g5_* and g7_* (captured from the reference); the numpy statements the native loops were written from are
test infrastructure (tests/conditioning_numpy.py).
"""
import ctypes as C
import warnings

import numpy as np
from scipy import ndimage


def _native():
    from . import _ffi
    return _ffi.load()


def _ptr(a):
    return a.ctypes.data_as(C.c_void_p)

_EIGHT = np.ones((3, 3), bool)
_CROSS = np.array([[0, 1, 0], [1, 1, 1], [0, 1, 0]])
_RING = np.array([[1, 1, 1], [1, 0, 1], [1, 1, 1]], bool)
_SQRT2 = np.sqrt(2.0)


def _sea_mask(a, below_sea):
    return (a != 0) if below_sea else (a > 0)


# ----------------------------------------------------------------------------------------------
# quantisation artefacts (reference :396-426)
# ----------------------------------------------------------------------------------------------
def fill_pit_artifacts(elev, maximum_pit_area=32.0, fill_flats_below_sea=False):
    """Raise by one unit every small flat depression whose whole 8-connected rim is exactly one unit
    higher -- the signature of integer quantisation.  Keeps the input dtype."""
    if np.ma.isMaskedArray(elev):
        elev = np.ma.filled(elev.astype(np.float64), np.nan)          # masked cells are no-data (:213-214)
    elev = np.asarray(elev)
    if elev.ndim != 2:
        raise ValueError("elevation must be a 2-D array")
    low = (ndimage.minimum_filter(elev, (3, 3)) >= elev) & _sea_mask(elev, fill_flats_below_sea)
    lab, nlab = ndimage.label(low, structure=_EIGHT)
    lab = np.ascontiguousarray(lab, np.int32)
    z = np.ascontiguousarray(elev, np.float64)
    up = np.empty(elev.shape, np.uint8)
    lib = _native()
    # (a float32 surface computes `rim - 1` in float32, :424: the library is told)
    if lib.pydem_cond_pit_artifacts(_ptr(z), elev.shape[0], elev.shape[1], _ptr(lab), int(nlab), float(maximum_pit_area),
                                    int(elev.dtype == np.float32), _ptr(up)):
        raise RuntimeError("pydem_cond_pit_artifacts failed")
    return elev + up.astype(elev.dtype)


# ----------------------------------------------------------------------------------------------
# flats (reference :551-579, :308-394)
# ----------------------------------------------------------------------------------------------
def fill_flats(elev, maximum_pit_area=32.0, fill_flats_below_sea=False, fill_flats_source_tol=1,
               fill_flats_peaks=True, fill_flats_pits=True):
    """calc_fill_flats (:551-579): artefact pits first (when maximum_pit_area is set), then every flat is
    re-surfaced between its uphill rim and its outlet.  Returns float64."""
    if maximum_pit_area:
        elev = fill_pit_artifacts(elev, maximum_pit_area, fill_flats_below_sea)
    data = np.ascontiguousarray(np.ma.filled(np.asarray(elev).astype('float64'), np.nan))
    built = data.copy()
    flat = (ndimage.minimum_filter(data, (3, 3)) >= data) & _sea_mask(data, fill_flats_below_sea)
    flat[0, 0] = flat[-1, 0] = flat[0, -1] = flat[-1, -1] = False          # corners never (:569-572)
    lab, nlab = ndimage.label(flat, structure=_EIGHT)
    lab = np.ascontiguousarray(lab, np.int32)
    lib = _native()
    if lib.pydem_cond_fill_flats(_ptr(data), _ptr(built), data.shape[0], data.shape[1], _ptr(lab), int(nlab),
                                 float(fill_flats_source_tol), int(bool(fill_flats_peaks)), int(bool(fill_flats_pits))):
        raise RuntimeError("pydem_cond_fill_flats failed")
    return built


# ----------------------------------------------------------------------------------------------
# pit drain paths (reference :428-548)
# ----------------------------------------------------------------------------------------------
def pit_drain_paths(elev, dX, dY, drain_pits_max_iter=300, drain_pits_max_dist=32, drain_pits_max_dist_XY=None,
                    fill_flats_below_sea=False):
    """calc_pit_drain_paths (:428-548): every strict local minimum, lowest first, is connected to the first
    lower cell found by growing a region through its lowest rim cells; the cells on the way get elevations
    that fall linearly from pit to outlet.  The surface is edited IN PLACE and sequentially -- later pits
    see earlier paths -- in the array's own dtype, exactly like the reference (integer surfaces truncate the
    path values, float32 surfaces round them).  Returns (elev, n_failed, max_iter_used)."""
    if np.ma.isMaskedArray(elev):
        elev = np.ma.filled(elev.astype(np.float64), np.nan)
    elev = np.asarray(elev)
    if elev.ndim != 2:
        raise ValueError("elevation must be a 2-D array")
    if not (elev.flags.c_contiguous and elev.flags.writeable):
        elev = np.array(elev, order='C')
    dtype = elev.dtype
    # the library works on float64 values; `mode` tells it the dtype whose rounding the stored path values get (:539)
    mode = 0 if dtype == np.float64 else (2 if dtype == np.float32 else (1 if dtype.kind in 'iu' else None))
    if mode is None:
        raise TypeError("unsupported elevation dtype %r" % (dtype,))
    e = elev.ravel()
    lows = (ndimage.minimum_filter(elev, footprint=_RING).ravel() > e) & _sea_mask(e, fill_flats_below_sea)
    pit_ids = np.where(lows)[0]
    # the reference's own call on the array's own dtype (:450): the tie order of numpy's sort is part of the result
    order = np.ascontiguousarray(pit_ids[np.argsort(e[pit_ids])], np.int64)
    work = elev if mode == 0 else np.ascontiguousarray(elev, np.float64)
    dXc = np.ascontiguousarray(dX, np.float64)
    dYc = np.ascontiguousarray(dY, np.float64)
    failed = C.c_int64(0)
    used = C.c_int64(0)
    lib = _native()
    if lib.pydem_cond_pit_paths(_ptr(work), elev.shape[0], elev.shape[1], _ptr(order), order.size, _ptr(dXc), dXc.size, _ptr(dYc),
                                int(drain_pits_max_iter), int(drain_pits_max_dist or 0),
                                float(drain_pits_max_dist_XY) if drain_pits_max_dist_XY else 0.0, int(mode), C.byref(failed), C.byref(used)):
        raise RuntimeError("pydem_cond_pit_paths failed")
    if mode != 0:
        elev[...] = work.astype(dtype)
    if failed.value:
        warnings.warn("Warning %d pits had no place to drain to in this chunk" % failed.value)
    return elev, int(failed.value), int(used.value)
