"""ctypes binding of libpydem_hip.so (C-ABI declared in include/pydem_hip.h).

The product path has no CPU fallback: if the library is missing or no MI355X is visible the
calls raise.  Nothing here imports torch or the oracle.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'lib', 'libpydem_hip.so')

# enum pydem_field
ELEV, MAG, DIRECTION, FLATS, SECTION, PROPORTION, UCA, TWI, EDGE_TODO, EDGE_DONE = range(10)
FIELD_DTYPE = {ELEV: np.float64, MAG: np.float64, DIRECTION: np.float64, FLATS: np.uint8,
               SECTION: np.int8, PROPORTION: np.float64, UCA: np.float64, TWI: np.float64,
               EDGE_TODO: np.uint8, EDGE_DONE: np.uint8}
# enum pydem_dtype
_DTYPES = {np.dtype('float64'): 0, np.dtype('float32'): 1, np.dtype('int16'): 2, np.dtype('int32'): 3,
           np.dtype('uint8'): 4, np.dtype('int8'): 5, np.dtype('bool'): 4}


class Options(C.Structure):
    """struct pydem_options (defaults = DEMProcessor traits, reference dem_processing.py:105-154)"""
    _fields_ = [('drain_pits', C.c_int32), ('drain_pits_min_border', C.c_int32),
                ('drain_pits_max_iter', C.c_int32), ('drain_pits_max_dist', C.c_int32),
                ('drain_pits_max_dist_XY', C.c_double), ('apply_uca_limit_edges', C.c_int32),
                ('apply_twi_limits', C.c_int32), ('apply_twi_limits_on_uca', C.c_int32),
                ('circular_ref_maxcount', C.c_int32), ('uca_saturation_limit', C.c_double),
                ('twi_min_slope', C.c_double), ('twi_min_area', C.c_double)]


class Timings(C.Structure):
    """struct pydem_timings"""
    _fields_ = [('slopes_directions_ms', C.c_double), ('stencil_kernel_ms', C.c_double),
                ('flats_ms', C.c_double), ('graph_ms', C.c_double), ('pits_ms', C.c_double),
                ('sweep_ms', C.c_double), ('twi_ms', C.c_double), ('sweep_rounds', C.c_int64),
                ('sweep_kernel_launches', C.c_int64), ('n_flats', C.c_int64), ('n_pit_edges', C.c_int64),
                ('n_pits_undrained', C.c_int64), ('n_unresolved', C.c_int64), ('sweep_tile_passes', C.c_int64),
                ('n_pits', C.c_int64), ('n_pits_row', C.c_int64), ('n_pits_wave', C.c_int64), ('n_pits_big', C.c_int64)]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_}


# every symbol include/pydem_hip.h declares: name -> (restype, argtypes)
_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
SYMBOLS = {
    'pydem_hip_last_error': (C.c_char_p, []),
    'pydem_hip_device_count': (C.c_int, [C.POINTER(C.c_int)]),
    'pydem_hip_device_memory': (C.c_int, [C.c_int, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    'pydem_hip_release_scratch': (C.c_int, []),
    'pydem_hip_device_name': (C.c_int, [C.c_int, C.c_char_p, C.c_int]),
    'pydem_tile_create': (C.c_int, [C.c_int64, C.c_int64, C.c_int, _PP]),
    'pydem_tile_destroy': (C.c_int, [_P]),
    'pydem_tile_set_spacing': (C.c_int, [_P, _P, _P, _P, _P]),
    'pydem_tile_upload': (C.c_int, [_P, C.c_int, _P, C.c_int]),
    'pydem_tile_download': (C.c_int, [_P, C.c_int, _P]),
    'pydem_tile_get_line': (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P]),
    'pydem_tile_set_line': (C.c_int, [_P, C.c_int, C.c_int, C.c_int64, _P]),
    'pydem_tile_get_lines': (C.c_int, [_P, C.c_int, _P, _P, _P, _P]),
    'pydem_cond_pit_artifacts': (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int32, C.c_double, C.c_int, _P]),
    'pydem_cond_fill_flats': (C.c_int, [_P, _P, C.c_int64, C.c_int64, _P, C.c_int32, C.c_double, C.c_int, C.c_int]),
    'pydem_cond_pit_paths': (C.c_int, [_P, C.c_int64, C.c_int64, _P, C.c_int64, _P, C.c_int64, _P, C.c_int, C.c_int,
                                       C.c_double, C.c_int, _P, _P]),
    'pydem_tiff_lzw_encode': (C.c_int, [_P, C.c_int64, _P, C.c_int64, _P]),
    'pydem_tile_synchronize': (C.c_int, [_P]),
    'pydem_tile_timings': (C.c_int, [_P, C.POINTER(Timings)]),
    'pydem_tile_device_bytes': (C.c_int64, [_P]),
    'pydem_tile_synth_fractal': (C.c_int, [_P, C.c_uint32, C.c_int64, C.c_int64, C.c_int, C.c_int,
                                           C.c_double, C.c_double]),
    'pydem_fill_flats': (C.c_int, [_P, C.c_double, C.c_int, C.c_double, C.c_int, C.c_int, C.c_int, _P]),
    'pydem_pit_candidates': (C.c_int, [_P, C.c_int, C.POINTER(C.c_int64)]),
    'pydem_pit_candidates_read': (C.c_int, [_P, C.c_int64, _P, _P]),
    'pydem_pit_paths': (C.c_int, [_P, _P, C.c_int64, C.c_int, C.c_int, C.c_double, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64), _P]),
    'pydem_slopes_directions': (C.c_int, [_P]),
    'pydem_find_flats': (C.c_int, [_P]),
    'pydem_uca': (C.c_int, [_P, C.POINTER(Options)]),
    'pydem_build_graph': (C.c_int, [_P, C.POINTER(Options)]),
    'pydem_uca_edge_update': (C.c_int, [_P, C.POINTER(Options), _PP, _PP, _PP]),
    'pydem_uca_edge_round_inc': (C.c_int, [_P, C.POINTER(Options), _PP, _PP, _PP]),
    'pydem_uca_edge_round_inc_dev': (C.c_int, [_P, C.POINTER(Options)]),
    'pydem_uca_edge_flush': (C.c_int, [_P]),
    'pydem_twi': (C.c_int, [_P, C.POINTER(Options)]),
    'pydem_tile_pit_edges': (C.c_int, [_P, C.POINTER(C.c_int64), _P, _P, _P]),
    'pydem_tile_graph_words': (C.c_int, [_P, _P]),
    'pydem_tile_restore_pit_slopes': (C.c_int, [_P]),
    'pydem_bench_stencil': (C.c_int, [_P, C.c_int, C.POINTER(C.c_double)]),
    'pydem_drain_area': (C.c_int, [_P, _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, _P, _P, C.c_int, C.c_int]),
    'pydem_drain_connections': (C.c_int, [_P, _P, _P, _P, C.c_int64, C.c_uint8, C.c_int]),
    'pydem_comm_unique_id': (C.c_int, [C.c_char_p]),
    'pydem_comm_create': (C.c_int, [C.c_int, C.c_int, C.c_char_p, C.c_int, _PP]),
    'pydem_comm_destroy': (C.c_int, [_P]),
    'pydem_comm_begin': (C.c_int, [_P, C.c_int64]),
    'pydem_comm_pack_line': (C.c_int, [_P, _P, C.c_int, C.c_int, C.c_int64, C.c_int64]),
    'pydem_comm_pack_lines': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    'pydem_comm_put': (C.c_int, [_P, _P, C.c_int64, C.c_int64]),
    'pydem_comm_allreduce': (C.c_int, [_P, C.c_int64, C.c_int, _P]),
    'pydem_board_create': (C.c_int, [C.c_int, C.c_int, C.c_int64, _PP]),
    'pydem_board_destroy': (C.c_int, [_P]),
    'pydem_board_set_desc': (C.c_int, [_P, C.c_int, C.c_int32, C.c_int32, _P, _P, _P]),
    'pydem_board_set_lines': (C.c_int, [_P, C.c_int, C.c_int64, C.c_int64, _P, C.c_int, _P, _P, _P, _P]),
    'pydem_board_refresh': (C.c_int, [_P, _P, C.c_int, _P]),
    'pydem_board_refresh_stage': (C.c_int, [_P, C.c_int, _P, _P, C.c_int64, _P]),
    'pydem_board_refresh_unstage': (C.c_int, [_P, C.c_int, _P, _P, C.c_int64]),
    'pydem_board_eval': (C.c_int, [_P, C.c_int, _P, _P, _P]),
    'pydem_board_download': (C.c_int, [_P, _P]),
    'pydem_board_run_waves': (C.c_int, [_P, _P, C.c_int, _P, _P]),
    'pydem_board_prepare_waves': (C.c_int, [_P, C.c_int, C.c_uint64]),
    'pydem_board_run_waves_ex': (C.c_int, [_P, _P, C.c_int, _P, _P, _P, _P]),
    'pydem_comm_count': (C.c_int, [_P, _P]),
    'pydem_tile_edge_queue_ready': (C.c_int, [_P]),
}

_lib = None


def load():
    """dlopen the library and bind every declared symbol (fails loudly if anything is missing)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError("libpydem_hip.so is not built (%s); run `python -m pydem_amd.build`. "
                               "There is no CPU fallback." % LIB_PATH)
        lib = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(lib, name)   # AttributeError if the export is missing
            fn.restype = res
            fn.argtypes = args
        _lib = lib
    return _lib


class HipError(RuntimeError):
    pass


def check(rc):
    if rc != 0:
        raise HipError("libpydem_hip error %d: %s" % (rc, load().pydem_hip_last_error().decode()))


def device_count():
    n = C.c_int(0)
    check(load().pydem_hip_device_count(C.byref(n)))
    return n.value


def release_scratch():
    """Return the per-device scratch arenas of the conditioning stages to the driver."""
    check(load().pydem_hip_release_scratch())


def device_memory(device=0):
    """(free, total) bytes of the device's HBM."""
    f, tot = C.c_int64(0), C.c_int64(0)
    check(load().pydem_hip_device_memory(int(device), C.byref(f), C.byref(tot)))
    return f.value, tot.value


class Tile(object):
    """Owns one device-resident tile handle."""

    def __init__(self, n_rows, n_cols, device=0):
        self.lib = load()
        self.shape = (int(n_rows), int(n_cols))
        self._h = C.c_void_p()
        check(self.lib.pydem_tile_create(n_rows, n_cols, device, C.byref(self._h)))

    def close(self):
        if self._h:
            self.lib.pydem_tile_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_spacing(self, dX, dY, dX2, dY2):
        arrs = [np.ascontiguousarray(a, np.float64) for a in (dX, dY, dX2, dY2)]
        n = self.shape[0]
        assert arrs[0].size == n - 1 and arrs[1].size == n - 1 and arrs[2].size == n and arrs[3].size == n
        check(self.lib.pydem_tile_set_spacing(self._h, *[a.ctypes.data_as(_P) for a in arrs]))

    def upload(self, field, arr):
        arr = np.ascontiguousarray(arr)
        if arr.shape != self.shape:
            raise ValueError("field %d: array of shape %r uploaded to a tile of shape %r" % (field, arr.shape, self.shape))
        if FIELD_DTYPE[field] != np.float64:
            arr = np.ascontiguousarray(arr.astype(FIELD_DTYPE[field], copy=False))
            dt = _DTYPES[arr.dtype]
        else:
            if arr.dtype not in _DTYPES or arr.dtype == np.bool_:
                arr = np.ascontiguousarray(arr, np.float64)
            dt = _DTYPES[arr.dtype]
        check(self.lib.pydem_tile_upload(self._h, field, arr.ctypes.data_as(_P), dt))

    def download(self, field):
        out = np.empty(self.shape, FIELD_DTYPE[field])
        check(self.lib.pydem_tile_download(self._h, field, out.ctypes.data_as(_P)))
        return out

    def get_line(self, field, axis, index):
        out = np.empty(self.shape[1] if axis == 0 else self.shape[0], FIELD_DTYPE[field])
        check(self.lib.pydem_tile_get_line(self._h, field, axis, index, out.ctypes.data_as(_P)))
        return out

    def get_lines(self, requests):
        """requests: [(field, axis, index)] -> list of 1-D arrays, one device synchronisation for all."""
        k = len(requests)
        outs = [np.empty(self.shape[1] if axis == 0 else self.shape[0], FIELD_DTYPE[field]) for field, axis, _ in requests]
        fields = (C.c_int * k)(*[r[0] for r in requests])
        axes = (C.c_int * k)(*[r[1] for r in requests])
        idx = (C.c_int64 * k)(*[r[2] for r in requests])
        dsts = (C.c_void_p * k)(*[o.ctypes.data for o in outs])
        check(self.lib.pydem_tile_get_lines(self._h, k, fields, axes, idx, dsts))
        return outs

    def set_line(self, field, axis, index, values):
        v = np.ascontiguousarray(values, FIELD_DTYPE[field])
        assert v.size == (self.shape[1] if axis == 0 else self.shape[0])
        check(self.lib.pydem_tile_set_line(self._h, field, axis, index, v.ctypes.data_as(_P)))

    def synth_fractal(self, seed=0, row0=0, col0=0, n_octaves=12, top_shift=12, zmin=1.0, zrange=1000.0):
        check(self.lib.pydem_tile_synth_fractal(self._h, seed, row0, col0, n_octaves, top_shift, zmin, zrange))

    def slopes_directions(self):
        check(self.lib.pydem_slopes_directions(self._h))

    def find_flats(self):
        check(self.lib.pydem_find_flats(self._h))

    def fill_flats(self, max_pit_area, below_sea, source_tol, peaks, pits, artefacts_only=False):
        """Conditioning on the resident elevation; returns False when the tile must go through the host path (NaN cells)."""
        flag = C.c_int(0)
        check(self.lib.pydem_fill_flats(self._h, float(max_pit_area or 0.0), int(bool(below_sea)), float(source_tol), int(bool(peaks)),
                                        int(bool(pits)), int(bool(artefacts_only)), C.byref(flag)))
        return flag.value == 0

    def pit_drain_paths(self, below_sea, max_iter, max_dist, max_dist_XY, sort_dtype=None):
        """calc_pit_drain_paths on the resident elevation.  Returns (n_failed, iterations used, rounds), or None when the tile
        must go through the host loop (no-data cells, float32 surface, or the parallel schedule gave up: surface restored)."""
        import time
        t0 = time.perf_counter()
        n = C.c_int64(0)
        check(self.lib.pydem_pit_candidates(self._h, int(bool(below_sea)), C.byref(n)))
        if n.value < 0:
            return None
        cells = np.empty(max(n.value, 1), np.int32); elev = np.empty(max(n.value, 1), np.float64)
        check(self.lib.pydem_pit_candidates_read(self._h, n.value, cells.ctypes.data_as(_P), elev.ctypes.data_as(_P)))
        cells, elev = cells[:n.value], elev[:n.value]
        t1 = time.perf_counter()
        # the reference's call (:450) on the array's own dtype: the tie order of numpy's sort is part of the result and differs between dtypes
        keys = elev if sort_dtype is None or np.dtype(sort_dtype) == np.float64 else elev.astype(sort_dtype)
        order = np.ascontiguousarray(cells[np.argsort(keys)], np.int32)
        t2 = time.perf_counter()
        failed, used, rounds, flag = C.c_int64(0), C.c_int64(0), C.c_int64(0), C.c_int(0)
        check(self.lib.pydem_pit_paths(self._h, order.ctypes.data_as(_P), n.value, int(max_iter), int(max_dist or 0),
                                       float(max_dist_XY) if max_dist_XY else 0.0, C.byref(failed), C.byref(used), C.byref(rounds),
                                       C.byref(flag)))
        if os.environ.get('PYDEM_PATHS_DEBUG'):
            import sys
            sys.stderr.write("pit paths host side: candidates %.1f ms, argsort %.1f ms, pydem_pit_paths %.1f ms\n"
                             % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, (time.perf_counter() - t2) * 1e3))
        if flag.value:
            return None
        return int(failed.value), int(used.value), int(rounds.value)

    def uca(self, opt):
        check(self.lib.pydem_uca(self._h, C.byref(opt)))

    def build_graph(self, opt):
        check(self.lib.pydem_build_graph(self._h, C.byref(opt)))

    def uca_edge_update(self, opt, data, done, todo, incremental=False):
        """data/done/todo: sequences (left, right, top, bottom) of 1-D arrays.  incremental: pydem_uca_edge_round_inc."""
        n, m = self.shape
        lens = (n, n, m, m)
        d = [np.ascontiguousarray(np.asarray(a, np.float64).ravel()) for a in data]
        dn = [np.ascontiguousarray(np.asarray(a).ravel().astype(np.uint8)) for a in done]
        td = [np.ascontiguousarray(np.asarray(a).ravel().astype(np.uint8)) for a in todo]
        for arrs in (d, dn, td):
            assert [a.size for a in arrs] == list(lens), "edge strips must have n_rows / n_cols entries"
        pack = lambda xs: (C.c_void_p * 4)(*[x.ctypes.data_as(C.c_void_p) for x in xs])
        fn = self.lib.pydem_uca_edge_round_inc if incremental else self.lib.pydem_uca_edge_update
        check(fn(self._h, C.byref(opt), pack(d), pack(dn), pack(td)))

    def uca_edge_flush(self):
        check(self.lib.pydem_uca_edge_flush(self._h))

    def edge_queue_ready(self):
        """Can the next rounds of this tile be queued behind a device-side wave selection (Board.run_waves)?"""
        return bool(self.lib.pydem_tile_edge_queue_ready(self._h))

    def uca_edge_round_dev(self, opt):
        """Incremental edge round with the strips the edge board wrote into the tile's buffers."""
        check(self.lib.pydem_uca_edge_round_inc_dev(self._h, C.byref(opt)))

    def twi(self, opt):
        check(self.lib.pydem_twi(self._h, C.byref(opt)))

    def graph_words(self):
        """The packed graph word of every cell (static bits: in-mask, out flags, pit flags, facet), uint32 [n, m]."""
        out = np.empty(self.shape, np.uint32)
        check(self.lib.pydem_tile_graph_words(self._h, out.ctypes.data_as(_P)))
        return out

    def pit_edges(self):
        """(pit, drain, weight) triplets of the last uca() call, in emission order."""
        n = C.c_int64(0)
        check(self.lib.pydem_tile_pit_edges(self._h, C.byref(n), None, None, None))
        src = np.empty(n.value, np.int32); dst = np.empty(n.value, np.int32); w = np.empty(n.value, np.float64)
        if n.value:
            check(self.lib.pydem_tile_pit_edges(self._h, C.byref(n), src.ctypes.data_as(_P), dst.ctypes.data_as(_P),
                                                w.ctypes.data_as(_P)))
        return src, dst, w

    def restore_pit_slopes(self):
        check(self.lib.pydem_tile_restore_pit_slopes(self._h))

    def bench_stencil(self, iters):
        ms = C.c_double(0)
        check(self.lib.pydem_bench_stencil(self._h, iters, C.byref(ms)))
        return ms.value

    def synchronize(self):
        check(self.lib.pydem_tile_synchronize(self._h))

    def timings(self):
        tm = Timings()
        check(self.lib.pydem_tile_timings(self._h, C.byref(tm)))
        return tm.as_dict()

    def device_bytes(self):
        return self.lib.pydem_tile_device_bytes(self._h)


class Board(object):
    """Device-resident edge board (include/pydem_hip.h, pydem_board_*)."""

    def __init__(self, device, n_tiles, n_doubles):
        self.lib = load()
        self.n_tiles, self.n_doubles = n_tiles, n_doubles
        self._h = C.c_void_p()
        check(self.lib.pydem_board_create(device, n_tiles, n_doubles, C.byref(self._h)))
        self._out = np.zeros((n_tiles, 8), np.uint64)

    def close(self):
        if self._h:
            self.lib.pydem_board_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_desc(self, index, n, m, offsets28, flags8, tile=None):
        o = np.ascontiguousarray(offsets28, np.int64); f = np.ascontiguousarray(flags8, np.int32)
        assert o.size == 28 and f.size == 8
        check(self.lib.pydem_board_set_desc(self._h, index, n, m, o.ctypes.data_as(_P), f.ctypes.data_as(_P),
                                            tile._h if tile is not None else None))

    def set_lines(self, index, mb_start, size, tile=None, lines=()):
        """lines: [(field, axis, index, rel_offset)] of `tile` (a tile of this process), or nothing for a remote tile."""
        k = len(lines)
        fields = (C.c_int * max(k, 1))(*[l[0] for l in lines]); axes = (C.c_int * max(k, 1))(*[l[1] for l in lines])
        idx = (C.c_int64 * max(k, 1))(*[l[2] for l in lines]); offs = (C.c_int64 * max(k, 1))(*[l[3] for l in lines])
        check(self.lib.pydem_board_set_lines(self._h, index, int(mb_start), int(size), tile._h if tile is not None else None,
                                             k, fields, axes, idx, offs))

    def refresh(self, comm, tiles, host_sum=None):
        """Replicate the lines of `tiles` on the board: pack, sum over ranks, scatter.  comm: an RCCL communicator (or None in
        a single process); host_sum: a function that sums a float64 array over all ranks in place, for transports without
        RCCL (the staging buffer then makes the round trip through host memory)."""
        tiles = list(tiles)
        for k0 in range(0, len(tiles), 64):
            part = tiles[k0:k0 + 64]
            arr = (C.c_int * len(part))(*part)
            if host_sum is None:
                check(self.lib.pydem_board_refresh(self._h, comm._h if comm is not None else None, len(part), arr))
                continue
            buf = np.empty(self.n_doubles, np.float64)
            n = C.c_int64(0)
            check(self.lib.pydem_board_refresh_stage(self._h, len(part), arr, buf.ctypes.data_as(_P), buf.size, C.byref(n)))
            part_buf = buf[:n.value]
            host_sum(part_buf)
            check(self.lib.pydem_board_refresh_unstage(self._h, len(part), arr, part_buf.ctypes.data_as(_P), n.value))

    def eval(self, tiles, full):
        k = len(tiles)
        t = (C.c_int * max(k, 1))(*tiles); f = (C.c_int * max(k, 1))(*[int(v) for v in full])
        check(self.lib.pydem_board_eval(self._h, k, t, f, self._out.ctypes.data_as(_P)))
        return self._out

    def download(self):
        out = np.empty(self.n_doubles, np.float64)
        check(self.lib.pydem_board_download(self._h, out.ctypes.data_as(_P)))
        return out

    # layout of the queued waves' state (csrc/comm.hip, SCH_*)
    SCH_OK, SCH_STOP, SCH_NWAVES, SCH_LIMIT, SCH_GRAPH, SCH_ND, SCH_PD, SCH_HASH, SCH_HAS, SCH_READERS, SCH_NBRS, SCH_LOG, SCH_NTB, SCH_TBLOG, SCH_WORDS = \
        0, 1, 2, 3, 7, 8, 72, 136, 200, 264, 328, 392, 521, 522, 528

    def prepare_waves(self, staged, ok_tiles):
        """Everything a batch of queued waves over the tiles `ok_tiles` (bit mask) needs that can fail on this rank alone
        (pydem_board_prepare_waves); nothing is enqueued.  Raises HipError; with several ranks the caller lets the ranks agree on
        the outcome before any of them calls run_waves."""
        check(self.lib.pydem_board_prepare_waves(self._h, 1 if staged else 0, C.c_uint64(int(ok_tiles))))

    _EXCHANGE = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int64)

    def run_waves(self, comm, k_waves, state, exchange=None):
        """Up to k_waves waves without a host look (pydem_board_run_waves); `state` (uint64[528]) is updated in place; returns the
        scalars of all tiles like eval().  `exchange` (instead of a communicator): an object with sum_bytes_inplace(uint8 array)
        and max_inplace(float64 array) that reduce over the ranks -- the staging buffer then goes through the host once per
        wave (pydem_board_run_waves_ex; the multi-process tests on one GPU)."""
        assert state.dtype == np.uint64 and state.size == self.SCH_WORDS and state.flags.c_contiguous
        if exchange is None:
            check(self.lib.pydem_board_run_waves(self._h, comm._h if comm is not None else None, int(k_waves), state.ctypes.data_as(_P),
                                                 self._out.ctypes.data_as(_P)))
            return self._out
        failure = []

        def cb(_ctx, op, buf, n):
            try:
                if op == 0:
                    exchange.sum_bytes_inplace(np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_uint8)), shape=(int(n),)))
                else:
                    exchange.max_inplace(np.ctypeslib.as_array(C.cast(buf, C.POINTER(C.c_double)), shape=(int(n),)))
                return 0
            except Exception as exc:          # (an exception must not unwind through the C frames)
                failure.append(exc)
                return 1
        fn = self._EXCHANGE(cb)
        rc = self.lib.pydem_board_run_waves_ex(self._h, None, int(k_waves), state.ctypes.data_as(_P), self._out.ctypes.data_as(_P), fn, None)
        if failure:
            raise failure[0]
        check(rc)
        return self._out


class Comm(object):
    """RCCL communicator handle (one per process / GPU)."""

    def __init__(self, world, rank, uid, device=0):
        self.lib = load()
        self.world, self.rank, self.device = world, rank, device
        self._h = C.c_void_p()
        check(self.lib.pydem_comm_create(world, rank, uid, device, C.byref(self._h)))

    @staticmethod
    def unique_id():
        buf = C.create_string_buffer(128)
        check(load().pydem_comm_unique_id(buf))
        return buf.raw

    def count(self):
        """Ranks behind the communicator as RCCL reports them (ncclCommCount)."""
        n = C.c_int(0)
        check(self.lib.pydem_comm_count(self._h, C.byref(n)))
        return int(n.value)

    def close(self):
        if self._h:
            self.lib.pydem_comm_destroy(self._h)
            self._h = C.c_void_p()

    def begin(self, n):
        check(self.lib.pydem_comm_begin(self._h, n))

    def pack_line(self, tile, field, axis, index, offset):
        check(self.lib.pydem_comm_pack_line(self._h, tile._h, field, axis, index, offset))

    def pack_lines(self, tile, lines):
        """lines: [(field, axis, index, offset)] of ONE tile; no host synchronisation (see comm.hip)."""
        k = len(lines)
        fields = (C.c_int * k)(*[l[0] for l in lines]); axes = (C.c_int * k)(*[l[1] for l in lines])
        idx = (C.c_int64 * k)(*[l[2] for l in lines]); offs = (C.c_int64 * k)(*[l[3] for l in lines])
        check(self.lib.pydem_comm_pack_lines(self._h, tile._h, k, fields, axes, idx, offs))

    def put(self, values, offset=0):
        v = np.ascontiguousarray(values, np.float64)
        check(self.lib.pydem_comm_put(self._h, v.ctypes.data_as(_P), v.size, offset))

    def allreduce(self, n, op=0):
        out = np.empty(n, np.float64)
        check(self.lib.pydem_comm_allreduce(self._h, n, op, out.ctypes.data_as(_P)))
        return out
