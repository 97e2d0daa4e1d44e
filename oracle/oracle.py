"""ctypes binding + driver for the CPU oracle (oracle/pydem_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package `pydem_amd` never imports this module.

`OracleDEM` strings the C stages together in the order of the reference's
DEMProcessor.calc_twi() call stack (dem_processing.py:1647 -> :682 -> :587 -> :864).
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_f64p = np.ctypeslib.ndpointer(np.float64, flags='C_CONTIGUOUS')
_u8p = np.ctypeslib.ndpointer(np.uint8, flags='C_CONTIGUOUS')
_i8p = np.ctypeslib.ndpointer(np.int8, flags='C_CONTIGUOUS')
_i32p = np.ctypeslib.ndpointer(np.int32, flags='C_CONTIGUOUS')
_i64p = np.ctypeslib.ndpointer(np.int64, flags='C_CONTIGUOUS')


def build(force=False):
    so = os.path.join(_HERE, 'libpydem_oracle.so')
    src = os.path.join(_HERE, 'pydem_oracle.c')
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(['make', '-s', '-C', _HERE])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = C.CDLL(build())
        L.oracle_free.argtypes = [C.c_void_p]
        L.oracle_set_elev_f32.argtypes = [C.c_int]
        L.oracle_slopes_directions.argtypes = [_f64p, C.c_int64, C.c_int64, _f64p, _f64p, _f64p, _f64p]
        L.oracle_flats_edges.argtypes = [_f64p, _f64p, _f64p, C.c_int64, C.c_int64, _u8p]
        L.oracle_section_proportion.argtypes = [_f64p, _u8p, C.c_int64, C.c_int64, _f64p, _f64p, _i8p, _f64p]
        L.oracle_pit_edges.restype = C.c_int64
        L.oracle_pit_edges.argtypes = [_f64p, _u8p, _f64p, C.c_int64, C.c_int64, _f64p, _f64p,
                                       C.c_int64, C.c_int64, C.c_double, C.c_int,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                       C.POINTER(C.c_int64)]
        L.oracle_adjacency.restype = C.c_int64
        L.oracle_adjacency.argtypes = [_i8p, _f64p, _f64p, C.c_int64, C.c_int64,
                                       _i64p, _i64p, _f64p, C.c_int64,
                                       C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        L.oracle_tocsr.argtypes = [_i32p, _i32p, C.c_int64, _i32p, _i32p]
        L.oracle_drain_area.restype = C.c_int64
        L.oracle_drain_area.argtypes = [_f64p, _u8p, _u8p, _i32p, _i32p, _f64p, _i32p, _i32p,
                                        C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_int]
        L.oracle_drain_connections.restype = C.c_int64
        L.oracle_drain_connections.argtypes = [_u8p, _u8p, _i32p, _i32p, C.c_int64, C.c_uint8]
        L.oracle_uca_chunk.argtypes = [_f64p, _i8p, _u8p, C.c_int64, C.c_int64, _f64p, _f64p,
                                       _i32p, _i32p, _f64p, C.c_int64, C.c_int, C.c_double,
                                       _f64p, _u8p, _u8p, _f64p]
        L.oracle_uca_update.argtypes = [_f64p, _u8p, C.c_int64, C.c_int64, _i32p, _i32p, _f64p,
                                        C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), C.POINTER(C.c_void_p),
                                        _f64p, _u8p, _u8p]
        L.oracle_twi.argtypes = [_f64p, _f64p, C.c_int64, C.c_double, C.c_double, C.c_double,
                                 C.c_int, C.c_int, _f64p]
        L.oracle_synth_fractal.argtypes = [_f64p, C.c_int64, C.c_int64, C.c_uint32, C.c_int64, C.c_int64,
                                           C.c_int, C.c_int, C.c_double, C.c_double]
        _LIB = L
    return _LIB


def _take(ptr, n, dtype):
    """Copy a malloc'ed C array into numpy and free it."""
    if n > 0:
        buf = (C.c_char * (n * np.dtype(dtype).itemsize)).from_address(ptr.value)
        out = np.frombuffer(buf, dtype=dtype, count=n).copy()
    else:
        out = np.zeros(0, dtype)
    lib().oracle_free(ptr)
    return out


def spacing_arrays(n_rows, dX=None, dY=None, dX2=None, dY2=None):
    """Scalar/array normalisation of DEMProcessor.__init__ (dem_processing.py:233-258)."""
    def norm(d, d2):
        if not isinstance(d, np.ndarray):
            val = 1 if d is None else d
            if d2 is None:
                d2 = np.ones(n_rows) * val
            d = np.ones(n_rows - 1) * val
        if d2 is None:
            d2 = np.ones(n_rows)
        return np.ascontiguousarray(d, np.float64), np.ascontiguousarray(d2, np.float64)
    dX, dX2 = norm(dX, dX2)
    dY, dY2 = norm(dY, dY2)
    return dX, dY, dX2, dY2


def slopes_directions(elev, dX, dY):
    n, m = elev.shape
    e = np.ascontiguousarray(elev, np.float64)
    mag = np.empty((n, m)); direction = np.empty((n, m))
    lib().oracle_set_elev_f32(int(np.asarray(elev).dtype == np.float32))     # float32 DEMs subtract in float32
    rc = lib().oracle_slopes_directions(e, n, m, dX, dY, mag, direction)
    lib().oracle_set_elev_f32(0)
    assert rc == 0
    return mag, direction


def flats_edges(elev, mag, direction):
    n, m = elev.shape
    flats = np.zeros((n, m), np.uint8)
    assert lib().oracle_flats_edges(np.ascontiguousarray(elev, np.float64), mag, direction, n, m, flats) == 0
    return flats


def section_proportion(direction, flats, dX, dY):
    n, m = direction.shape
    section = np.empty((n, m), np.int8); proportion = np.empty((n, m))
    assert lib().oracle_section_proportion(direction, np.ascontiguousarray(flats, np.uint8), n, m, dX, dY,
                                           section, proportion) == 0
    return section, proportion


def pit_edges(elev, flats, mag, dX, dY, max_iter=300, max_dist=32, max_dist_XY=None, min_border=False):
    """flats (uint8) and mag are patched in place like the reference does."""
    n, m = elev.shape
    pi, pj, pp = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nwarn = C.c_int64(0)
    lib().oracle_set_elev_f32(int(np.asarray(elev).dtype == np.float32))
    cnt = lib().oracle_pit_edges(np.ascontiguousarray(elev, np.float64), flats, mag, n, m, dX, dY,
                                 max_iter, max_dist or 0,
                                 float('nan') if not max_dist_XY else float(max_dist_XY), int(min_border),
                                 C.byref(pi), C.byref(pj), C.byref(pp), C.byref(nwarn))
    lib().oracle_set_elev_f32(0)
    assert cnt >= 0
    return _take(pi, cnt, np.int64), _take(pj, cnt, np.int64), _take(pp, cnt, np.float64), nwarn.value


def adjacency(section, proportion, elev, pit_i=None, pit_j=None, pit_prop=None):
    n, m = section.shape
    if pit_i is None:
        pit_i = np.zeros(0, np.int64); pit_j = np.zeros(0, np.int64); pit_prop = np.zeros(0, np.float64)
    a, b, c = C.c_void_p(), C.c_void_p(), C.c_void_p()
    nnz = lib().oracle_adjacency(section, proportion, np.ascontiguousarray(elev, np.float64), n, m,
                                 pit_i, pit_j, pit_prop, pit_i.size, C.byref(a), C.byref(b), C.byref(c))
    assert nnz >= 0
    indptr = _take(a, n * m + 1, np.int32)
    # indices/data were allocated with the pre-dedup size; only nnz entries are meaningful
    indices = _take(b, nnz + 1, np.int32)[:nnz].copy()
    data = _take(c, nnz + 1, np.float64)[:nnz].copy()
    return indptr, indices, data


def tocsr(indptr, indices, NN):
    rp = np.empty(NN + 1, np.int32); ri = np.empty(max(indices.size, 1), np.int32)
    lib().oracle_tocsr(indptr, np.ascontiguousarray(indices), NN, rp, ri)
    return rp, ri[:indices.size]


def drain_area(area, done, ids, col_indptr, col_indices, col_data, row_indptr, row_indices,
               n_rows, n_cols, edge_todo=None, edge_todo_no_mask=None, skip_edge=0):
    """Same signature/in-place behaviour as the reference's cyutils.drain_area (cyutils.pyx:78-116)."""
    et = edge_todo.ctypes.data_as(C.c_void_p) if edge_todo is not None else None
    etn = edge_todo_no_mask.ctypes.data_as(C.c_void_p) if edge_todo_no_mask is not None else None
    lib().oracle_drain_area(area, done.view(np.uint8), ids.view(np.uint8), col_indptr, col_indices, col_data,
                            row_indptr, row_indices, n_rows, n_cols, et, etn, int(skip_edge))
    return area, done, edge_todo, edge_todo_no_mask


def drain_connections(arr, ids, indptr, indices, set_to=0):
    lib().oracle_drain_connections(arr.view(np.uint8), ids.view(np.uint8), indptr, indices, arr.size, int(set_to))
    return arr


def uca_update(elev, flats, A, strips_data, strips_done, strips_todo, uca_init):
    """One edge-resolution round (oracle_uca_update).  strips_*: dicts keyed left/right/top/bottom.
    Returns (uca, edge_todo, edge_done)."""
    n, m = elev.shape
    keys = ('left', 'right', 'top', 'bottom')
    d = [np.ascontiguousarray(strips_data[k], np.float64).ravel() for k in keys]
    dn = [np.ascontiguousarray(strips_done[k]).astype(np.uint8).ravel() for k in keys]
    td = [np.ascontiguousarray(strips_todo[k]).astype(np.uint8).ravel() for k in keys]
    arr = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data_as(C.c_void_p) for x in xs])
    uca = np.ascontiguousarray(uca_init, np.float64).copy()
    todo = np.empty((n, m), np.uint8); done = np.empty((n, m), np.uint8)
    indptr, indices, data = A
    idx = indices if indices.size else np.zeros(1, np.int32)
    dat = data if data.size else np.zeros(1)
    rc = lib().oracle_uca_update(np.ascontiguousarray(elev, np.float64), np.ascontiguousarray(flats, np.uint8), n, m,
                                 indptr, idx, dat, arr(d), arr(dn), arr(td), uca, todo, done)
    assert rc == 0
    return uca, todo.astype(bool), done.astype(bool)


def twi(uca, mag, twi_min_slope=1e-3, twi_min_area=np.inf, uca_saturation_limit=32.0,
        apply_twi_limits=False, apply_twi_limits_on_uca=False):
    out = np.empty(uca.shape)
    lib().oracle_twi(np.ascontiguousarray(uca), np.ascontiguousarray(mag), uca.size, twi_min_slope, twi_min_area,
                     uca_saturation_limit, int(apply_twi_limits), int(apply_twi_limits_on_uca), out)
    return out


def synth_fractal(n, m, seed=0, row0=0, col0=0, n_octaves=12, top_shift=12, zmin=1.0, zrange=1000.0):
    z = np.empty((n, m))
    lib().oracle_synth_fractal(z, n, m, seed, row0, col0, n_octaves, top_shift, zmin, zrange)
    return z


class OracleDEM(object):
    """CPU restatement of DEMProcessor for the options on the hot path (no conditioning yet:
    fill_flats / drain_pits_path must be handled by the caller)."""

    def __init__(self, elev, dX=None, dY=None, dX2=None, dY2=None, drain_pits=True,
                 drain_pits_max_iter=300, drain_pits_max_dist=32, drain_pits_max_dist_XY=None,
                 drain_pits_min_border=False, circular_ref_maxcount=50, apply_uca_limit_edges=False,
                 uca_saturation_limit=32.0, twi_min_slope=1e-3, apply_twi_limits=False,
                 apply_twi_limits_on_uca=False):
        self.elev_in = np.asarray(elev)               # keeps the dtype: float32 DEMs subtract in float32
        self.elev = np.ascontiguousarray(elev, np.float64)
        self.dX, self.dY, self.dX2, self.dY2 = spacing_arrays(elev.shape[0], dX, dY, dX2, dY2)
        self.opt = dict(drain_pits=drain_pits, max_iter=drain_pits_max_iter, max_dist=drain_pits_max_dist,
                        max_dist_XY=drain_pits_max_dist_XY, min_border=drain_pits_min_border,
                        circ=circular_ref_maxcount, lim_edges=apply_uca_limit_edges,
                        sat=uca_saturation_limit, min_slope=twi_min_slope, twi_lim=apply_twi_limits,
                        twi_lim_uca=apply_twi_limits_on_uca)
        self.twi_min_area = np.inf
        self.mag = self.direction = self.flats = self.uca = None

    def calc_slopes_directions(self):
        self.mag, self.direction = slopes_directions(self.elev_in, self.dX, self.dY)
        self.mag_raw, self.direction_raw = self.mag.copy(), self.direction.copy()
        self.flats = flats_edges(self.elev, self.mag, self.direction)
        return self.mag, self.direction

    def build_graph(self):
        o = self.opt
        self.section, self.proportion = section_proportion(self.direction, self.flats, self.dX, self.dY)
        if o['drain_pits']:
            self.pit_i, self.pit_j, self.pit_prop, self.n_warn = pit_edges(
                self.elev_in, self.flats, self.mag, self.dX, self.dY, o['max_iter'], o['max_dist'],
                o['max_dist_XY'], o['min_border'])
        else:
            self.pit_i = self.pit_j = self.pit_prop = None
        self.A = adjacency(self.section, self.proportion, self.elev, self.pit_i, self.pit_j, self.pit_prop)

    def calc_uca(self):
        if self.direction is None:
            self.calc_slopes_directions()
        self.build_graph()
        o = self.opt
        n, m = self.elev.shape
        uca = np.empty((n, m)); todo = np.empty((n, m), np.uint8); done = np.empty((n, m), np.uint8)
        self.stats = np.zeros(4)
        indptr, indices, data = self.A
        idx = indices if indices.size else np.zeros(1, np.int32)
        dat = data if data.size else np.zeros(1)
        lib().oracle_uca_chunk(self.elev, self.section, self.flats, n, m, self.dX2, self.dY2,
                               indptr, idx, dat, o['circ'], int(o['lim_edges']), o['sat'],
                               uca, todo, done, self.stats)
        self.twi_min_area = min(self.twi_min_area, self.stats[3])
        self.uca, self.edge_todo, self.edge_done = uca, todo.astype(bool), done.astype(bool)
        return self.uca

    def calc_twi(self):
        if self.uca is None:
            self.calc_uca()
        o = self.opt
        t = twi(self.uca, self.mag, o['min_slope'], self.twi_min_area, o['sat'], o['twi_lim'], o['twi_lim_uca'])
        self.twi = t * 10
        return t
