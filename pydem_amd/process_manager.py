"""Directory orchestration: the `ProcessManager` drop-in for the accelerated path.

Mirrors the public surface of the reference class (creare-com/pydem v1.2.1,
pydem/process_manager.py:393-1318) for the part SURVEY.md section 8 puts in scope: tile grid
discovery (compute_grid :517-565), overlap / edge-line bookkeeping (compute_grid_overlaps
:601-740), the four phases (process_elevation :993, process_aspect_slope :1010, process_uca :1032,
process_uca_edges :1090), TWI (:1290-1317) and the stitched non-overlap arrays
(save_non_overlap_data :742-766), GeoTIFF export with 'average' overviews (:862-931) and the overview pyramid
(process_overviews :933-991).

What is different by design:
  * tiles stay resident on their GPU between phases (the reference round-trips every array
    through an on-disk zarr store and rebuilds a DEMProcessor -- and the whole flow graph -- per
    phase and per edge round, process_manager.py:54-315);
  * tiles shard round-robin over the visible GPUs (or over the ranks of a multi-process job),
    replacing multiprocessing.Pool; the only cross-tile coupling, the UCA edge fix-up, moves
    KiB-sized edge strips between neighbours (`EdgeTransport`);
  * the edge fix-up has the reference's two schedules: `edge_mode='reference'` is its serial loop
    (n_workers == 1, :1140-1211: one tile at a time in metric order -- the order the pm_* goldens were
    captured in), `edge_mode='pool'` is its multi-worker schedule (:1214-1246) as deterministic waves:
    the best-ranked tiles with finished neighbour edges run their edge rounds concurrently (one per
    GPU) from ONE snapshot of the strips, then the metrics of those tiles and their neighbours are
    refreshed.  See `process_uca_edges`.
"""
import logging
import os
import threading
import time

import numpy as np

from .dem_processing import DEMProcessor

logger = logging.getLogger(__name__)

EDGE_SLICES = {                                   # reference :41-50
    'left': (slice(0, None), 0),
    'right': (slice(0, None), -1),
    'top': (0, slice(0, None)),
    'bottom': (-1, slice(0, None)),
    'top-left': (0, 0),
    'top-right': (0, -1),
    'bottom-right': (-1, -1),
    'bottom-left': (-1, 0),
}
SIDES = ('left', 'right', 'top', 'bottom')
CORNERS = ('top-left', 'top-right', 'bottom-left', 'bottom-right')

DEBUG = False     # like the reference's module global (:52): force dX = dY = 1 inside the phases

DEM_PROC_KWARGS = (                               # whitelist of the reference (:401-433)
    "fill_flats", "fill_flats_below_sea", "fill_flats_source_tol", "fill_flats_peaks", "fill_flats_pits",
    "fill_flats_max_iter", "drain_pits", "drain_pits_path", "drain_pits_min_border", "drain_pits_spill",
    "drain_flats", "drain_pits_max_iter", "drain_pits_max_dist", "drain_pits_max_dist_XY",
    "apply_uca_limit_edges", "apply_twi_limits", "apply_twi_limits_on_uca", "uca_saturation_limit",
    "twi_min_slope", "twi_min_area", "circular_ref_maxcount", "maximum_pit_area")


def read_tile(fn):
    """Load one elevation tile: dict(elev, bounds=(left, bottom, right, top), dlon, dlat, dX.. optional).
    `.npz` tiles carry `elev` and `bounds`; optional `dX, dY, dX2, dY2` override the spacing."""
    if isinstance(fn, dict):
        # in-memory spec: {'elev': array | 'shape': (n, m) + 'synth': generator kwargs, 'bounds': (l, b, r, t)}
        out = dict(fn)
        shape = out['elev'].shape if out.get('elev') is not None else tuple(out['shape'])
        out['shape'] = shape
        left, bottom, right, top = [float(v) for v in out['bounds']]
        out.setdefault('dlon', (right - left) / shape[1])
        out.setdefault('dlat', -(top - bottom) / shape[0])
        return out
    ext = os.path.splitext(fn)[1].lower()
    if ext == '.npz':
        with np.load(fn) as d:
            out = {k: d[k] for k in d.files}
        elev = out['elev']
        left, bottom, right, top = [float(v) for v in out['bounds']]
        out['bounds'] = (left, bottom, right, top)
        out.setdefault('dlon', (right - left) / elev.shape[1])
        out.setdefault('dlat', -(top - bottom) / elev.shape[0])
        out['shape'] = elev.shape
        return out
    if ext in ('.tif', '.tiff'):
        # what the reference's workers get from rasterio + geopy (utils.dem_processor_from_raster_kwargs :46-51)
        from . import raster
        out = raster.dem_processor_from_raster_kwargs(fn)
        elev = out['elev']
        out['dlon'] = out['transform'][0]
        out['dlat'] = out['transform'][4]
        out['shape'] = elev.shape
        return out
    raise NotImplementedError("unsupported tile format %r (.npz tiles with `elev` and `bounds`, or GeoTIFF)" % ext)


class EdgeTransport(object):
    """Moves edge strips between tiles.  This default keeps everything in one process: tiles on any
    of the visible GPUs, strips staged through host memory (they are KiB-sized; the exchange is
    latency-bound).  `pydem_amd.parallel.DistTransport` carries the same strips between ranks."""

    def __init__(self, pm):
        self.pm = pm

    def owns(self, i):
        return True

    def gather_lines(self, requests):
        """requests: list of (tile, name, axis, index) -> list of 1-D arrays (None for missing tiles)."""
        out = [None] * len(requests)
        by_tile = {}
        for k, (t, name, axis, index) in enumerate(requests):
            if t >= 0:
                by_tile.setdefault(t, []).append(k)
        for t, ks in by_tile.items():
            dp = self.pm.tiles[t]
            reqs = [requests[k][1:] for k in ks]
            got = dp.get_lines(reqs) if hasattr(dp, 'get_lines') else [dp.get_line(*r) for r in reqs]
            for k, v in zip(ks, got):
                out[k] = v
        return out

    def allreduce_max(self, value):
        return value


class ProcessManager(object):
    dtype = np.float64
    grid_round_decimals = 2
    n_workers = 1
    in_path = '.'
    out_format = 'npy'
    _INPUT_FILE_TYPES = ["tif", "tiff", "npz"]

    def __init__(self, **kwargs):
        self.dem_proc_kwargs = {}
        self.devices = None
        self.processor_cls = DEMProcessor
        self.elev_conditioned = False      # inputs already conditioned: skip fill_flats / pit paths (see process_elevation)
        self.transport = None
        self.max_edge_rounds = 10000
        self.keep_first_pass_uca = True    # keep 'uca' (first pass) and 'uca_edges' separately like the reference's store
        self.checkpoint = False            # write every phase's per-tile results to `out_path` (out_format 'npy') and resume from them
        self.tiles_in_flight = None        # tiles of this process worked on at once in the per-tile phases (threads; the
                                           # library is re-entrant per tile handle, every tile has its own HIP stream).
                                           # None: one per GPU this process drives (1 for CPU-side processors)
        for k, v in kwargs.items():
            if k == 'dem_proc_kwargs':
                bad = [kk for kk in v if kk not in DEM_PROC_KWARGS]
                if bad:
                    raise ValueError("dem_proc_kwargs: unknown keys %r" % bad)
            setattr(self, k, v)
        if 'out_path' not in kwargs:
            self.out_path = os.path.join(self.in_path, 'results')
        if 'elev_source_files' not in kwargs:
            esf = [os.path.join(self.in_path, fn) for fn in os.listdir(self.in_path)
                   if os.path.splitext(fn)[-1].replace('.', '') in self._INPUT_FILE_TYPES]
            esf.sort()                                                      # reference :462-469
            self.elev_source_files = esf
        self._tile_meta = [read_tile(fn) for fn in self.elev_source_files]
        self.index = self.compute_index()
        self.tiles = [None] * self.n_inputs
        self.uca0 = [None] * self.n_inputs        # first-pass UCA per tile (the reference's 'uca' store)
        self.edge_rounds = 0
        if self.transport is None:
            self.transport = EdgeTransport(self)

    # ------------------------------------------------------------------ grid bookkeeping
    def _i(self, name):
        return ['left', 'bottom', 'right', 'top', 'dlon', 'dlat', 'nrows', 'ncols'].index(name)

    @property
    def n_inputs(self):
        return len(self.elev_source_files)

    def compute_index(self):
        """left, bottom, right, top, dlon, dlat, nrows, ncols per tile (reference :483-492)."""
        index = np.zeros((len(self.elev_source_files), 8))
        for i, meta in enumerate(self._tile_meta):
            index[i, :4] = np.array(meta['bounds'])
            index[i, 4] = meta['dlon']
            index[i, 5] = meta['dlat']
            index[i, 6:] = meta['shape']
        return index

    def compute_grid(self):
        """Tile -> (row, col) of a grid keyed by rounded top latitude / left longitude, and the side-by-side index space in
        which every grid row / column is as high / wide as its tiles (reference :517-565).  Whole-mosaic array
        expressions; the results are pinned against the reference's own bookkeeping (tests/test_process_manager_grid.py)."""
        I = self._i
        tops = np.round(self.index[:, I('top')], self.grid_round_decimals)
        lefts = np.round(self.index[:, I('left')], self.grid_round_decimals)
        ulat, row = np.unique(-tops, return_inverse=True)          # north first
        ulon, col = np.unique(lefts, return_inverse=True)
        n_rows, n_cols = ulat.size, ulon.size
        ids = np.arange(self.n_inputs)
        self.grid_shape = (n_rows, n_cols)
        self.grid_id = np.stack([row, col, col + row * n_cols], axis=1).astype(int)
        self.grid_id2i = -np.ones(self.grid_shape, dtype=int)
        self.grid_id2i[row, col] = ids
        heights = self.index[:, I('nrows')].astype(int)
        widths = self.index[:, I('ncols')].astype(int)

        def per_line(line, extent, count, what):
            lo = np.full(count, np.iinfo(int).max); hi = np.zeros(count, int)
            np.minimum.at(lo, line, extent); np.maximum.at(hi, line, extent)
            if np.any(lo != hi):
                raise AssertionError("tiles of grid %s %d differ in %s" % (what, int(np.nonzero(lo != hi)[0][0]), 'height' if what == 'row' else 'width'))
            return hi
        self.grid_lat_size = per_line(row, heights, n_rows, 'row')
        self.grid_lon_size = per_line(col, widths, n_cols, 'column')
        self._lat_cum = np.concatenate([[0], np.cumsum(self.grid_lat_size)]).astype(int)
        self._lon_cum = np.concatenate([[0], np.cumsum(self.grid_lon_size)]).astype(int)
        self.grid_slice = [(slice(int(self._lat_cum[r]), int(self._lat_cum[r + 1])), slice(int(self._lon_cum[c]), int(self._lon_cum[c + 1])))
                           for r, c in zip(row, col)]
        self.grid_size_tot = [int(self.grid_lat_size.sum()), int(self.grid_lon_size.sum())]
        self.grid_chunk = [int(self.grid_lat_size.min()), int(self.grid_lon_size.min())]

    @staticmethod
    def _calc_overlap(a, da, b, db, s, tie):
        """Pixels of overlap with a neighbour: for the non-overlap slices (first value) and for locating the neighbour's
        coincident edge line (second value, at least 1) -- reference :567-599.  Works on arrays."""
        gap = np.asarray(b, float) - np.asarray(a, float)
        mine = np.round((gap / da + tie - 0.01) / 2).astype(int)
        theirs = np.maximum(np.round(gap / db).astype(int), 1)
        return mine, theirs

    def compute_grid_overlaps(self):
        """Per tile: the part that is uniquely its own, and where (in the side-by-side index space) the neighbouring edge
        lines and corner pixels live (reference :601-740).  One array expression per side for the whole mosaic."""
        I = self._i
        idx = self.index
        row, col = self.grid_id[:, 0], self.grid_id[:, 1]
        n_rows, n_cols = self.grid_id2i.shape
        padded = -np.ones((n_rows + 2, n_cols + 2), int)
        padded[1:-1, 1:-1] = self.grid_id2i
        west, east = padded[row + 1, col], padded[row + 1, col + 2]
        north, south = padded[row, col + 1], padded[row + 2, col + 1]

        def side(nb, a_col, a_from_nb, d_col, b_col, b_from_nb, tie):
            """overlap with the neighbours `nb` (-1: none) along one axis; a / b pick the two coordinates to compare"""
            has = nb >= 0
            j = np.where(has, nb, 0)
            a = np.where(a_from_nb, idx[j, I(a_col)], idx[:, I(a_col)])
            b = np.where(b_from_nb, idx[j, I(b_col)], idx[:, I(b_col)])
            mine, theirs = self._calc_overlap(a, idx[:, I(d_col)], b, idx[j, I(d_col)], 0, tie)
            return np.where(has, mine, 0), np.where(has, theirs, 0)
        yes, no = np.ones(self.n_inputs, bool), np.zeros(self.n_inputs, bool)
        w_own, w_nb = side(west, 'left', no, 'dlon', 'right', yes, 0)
        e_own, e_nb = side(east, 'left', yes, 'dlon', 'right', no, 1)
        n_own, n_nb = side(north, 'top', no, 'dlat', 'bottom', yes, 0)
        s_own, s_nb = side(south, 'top', yes, 'dlat', 'bottom', no, 1)
        r0 = self._lat_cum[row]; r1 = self._lat_cum[row + 1]
        c0 = self._lon_cum[col]; c1 = self._lon_cum[col + 1]
        # corners exist wherever the grid continues in both directions, whether or not the diagonal tile does (:640-643)
        has_w, has_e, has_n, has_s = col > 0, col < n_cols - 1, row > 0, row < n_rows - 1
        self.grid_slice_unique, self.edge_data = [], []
        for i in range(self.n_inputs):
            rows, cols = self.grid_slice[i]
            self.grid_slice_unique.append((slice(int(r0[i] + n_own[i]), int(r1[i] - s_own[i])), slice(int(c0[i] + w_own[i]), int(c1[i] - e_own[i]))))
            top_line, bottom_line = int(r0[i] - n_nb[i]), int(r1[i] + s_nb[i] - 1)
            left_line, right_line = int(c0[i] - w_nb[i]), int(c1[i] + e_nb[i] - 1)
            corner = lambda tb, lr: (int((r0[i] - n_nb[i] * (has_n[i] and (has_w[i] if lr == 'l' else has_e[i]))) if tb == 't'
                                        else (r1[i] + s_nb[i] * (has_s[i] and (has_w[i] if lr == 'l' else has_e[i])) - 1)),
                                     int((c0[i] - w_nb[i] * (has_w[i] and (has_n[i] if tb == 't' else has_s[i]))) if lr == 'l'
                                        else (c1[i] + e_nb[i] * (has_e[i] and (has_n[i] if tb == 't' else has_s[i])) - 1)))
            self.edge_data.append({'left': (rows, left_line), 'right': (rows, right_line), 'top': (top_line, cols), 'bottom': (bottom_line, cols),
                                   'top-left': corner('t', 'l'), 'top-right': corner('t', 'r'),
                                   'bottom-left': corner('b', 'l'), 'bottom-right': corner('b', 'r')})
        # the mosaic without overlaps: every grid row / column as high / wide as the smallest unique part in it (:707-740)
        uh = np.array([s[0].stop - s[0].start for s in self.grid_slice_unique])
        uw = np.array([s[1].stop - s[1].start for s in self.grid_slice_unique])
        self.grid_lat_size_unique = np.full(n_rows, np.iinfo(int).max); np.minimum.at(self.grid_lat_size_unique, row, uh)
        self.grid_lon_size_unique = np.full(n_cols, np.iinfo(int).max); np.minimum.at(self.grid_lon_size_unique, col, uw)
        lat_cum = np.concatenate([[0], np.cumsum(self.grid_lat_size_unique)]).astype(int)
        lon_cum = np.concatenate([[0], np.cumsum(self.grid_lon_size_unique)]).astype(int)
        self.grid_slice_noverlap = [(slice(int(lat_cum[r]), int(lat_cum[r] + h)), slice(int(lon_cum[c]), int(lon_cum[c] + w)))
                                    for r, c, h, w in zip(row, col, uh, uw)]
        self.grid_size_tot_unique = [int(self.grid_lat_size_unique.sum()), int(self.grid_lon_size_unique.sum())]

    # ------------------------------------------------------------------ locating neighbour lines
    def _locate(self, grow, gcol):
        """Global (row, col) of the side-by-side index space -> (tile, local row, local col); tile = -1
        where the mosaic has a hole (the reference reads zeros/False from its zero-filled store)."""
        r = int(np.searchsorted(self._lat_cum, grow, side='right') - 1)
        c = int(np.searchsorted(self._lon_cum, gcol, side='right') - 1)
        if r < 0 or c < 0 or r >= self.grid_id2i.shape[0] or c >= self.grid_id2i.shape[1]:
            return -1, 0, 0
        return int(self.grid_id2i[r, c]), int(grow - self._lat_cum[r]), int(gcol - self._lon_cum[c])

    def _edge_line(self, i, key):
        """Where tile i's `key` edge data comes from: (tile, axis, local index) for the four sides (the
        line spans the full neighbour edge) or (tile, local row, local col) for the corners."""
        memo = self.__dict__.setdefault('_edge_line_memo', {})
        hit = memo.get((i, key))
        if hit is None:
            hit = memo[(i, key)] = self._edge_line_uncached(i, key)
        return hit

    def _edge_line_uncached(self, i, key):
        ed = self.edge_data[i][key]
        slc = self.grid_slice[i]
        if key in ('left', 'right'):
            t, _, lc = self._locate(slc[0].start, ed[1])
            return t, 1, lc
        if key in ('top', 'bottom'):
            t, lr, _ = self._locate(ed[0], slc[1].start)
            return t, 0, lr
        return self._locate(ed[0], ed[1])

    @staticmethod
    def check_1overlap(out_slice, edge_slc):
        """True when the neighbour line sits exactly one pixel outside the tile (reference :286-293)."""
        e = [getattr(v, 'start', v) for v in edge_slc]
        e_c = [min(max(e[0], out_slice[0].start), out_slice[0].stop - 1),
               min(max(e[1], out_slice[1].start), out_slice[1].stop - 1)]
        return any(abs(a - b) == 1 for a, b in zip(e_c, e))

    # ------------------------------------------------------------------ on-disk tile store and resume
    # The reference keeps every intermediate array in a zarr store under `out_path` together with a boolean table
    # success[n_tiles, 4] (elevation, aspect / slope, uca, twi; :998-1007, :1027-1029, :1057-1058, :1316-1317); a phase skips
    # the tiles whose flag is set, so a directory job that died continues where it stopped.  Tiles are resident here, so the
    # store is optional (`checkpoint=True`): one `.npy` file per tile and field (the reference's field names) plus
    # success.npy.  A resumed tile gets its stored fields back; the flow graph is rebuilt by the library the first time an
    # edge round needs it (process_manager.calc_uca_ec does the same from the stored elev / aspect / slope, :227-240).
    _SUCCESS_COLS = {'elev': 0, 'aspect_slope': 1, 'uca': 2, 'twi': 3}

    def _store_fn(self, i, key):
        return os.path.join(self.out_path, 'tile_%04d_%s.npy' % (i, key))

    def _store_lock(self):
        # _store() runs in the per-tile worker threads (tiles_in_flight > 1): one lock per manager for the table and its file
        lk = self.__dict__.get('_success_lock')
        if lk is None:
            lk = self.__dict__.setdefault('_success_lock', threading.RLock())
        return lk

    def _success(self):
        with self._store_lock():
            if getattr(self, '_success_table', None) is None:
                fn = os.path.join(self.out_path, 'success.npy')
                if self.checkpoint and os.path.exists(fn):
                    tab = np.load(fn)
                    if tab.shape != (self.n_inputs, 4):
                        raise ValueError("%s belongs to another mosaic" % fn)
                    self._success_table = tab.astype(bool)
                else:
                    self._success_table = np.zeros((self.n_inputs, 4), bool)
            return self._success_table

    def _store(self, i, phase, fields):
        """Write the fields of one finished tile, then its success flag.  Safe with several worker threads and with several
        ranks sharing one `out_path`: temporary names are unique per process and thread, and the table that is published
        is the OR of what is on disk and what this process knows, merged under a file lock (a rank only ever sets flags
        of its own tiles, so the last writer must not erase the others')."""
        if not self.checkpoint:
            return
        if self.out_format != 'npy':
            raise NotImplementedError("out_format %r (only 'npy' tile stores are written)" % (self.out_format,))
        os.makedirs(self.out_path, exist_ok=True)
        uniq = '.%d.%d.tmp.npy' % (os.getpid(), threading.get_ident())
        for key, arr in fields.items():
            tmp = self._store_fn(i, key) + uniq
            np.save(tmp, np.asarray(arr))
            os.replace(tmp, self._store_fn(i, key))
        if phase is not None:
            import fcntl
            fn = os.path.join(self.out_path, 'success.npy')
            with self._store_lock():
                tab = self._success()
                tab[i, self._SUCCESS_COLS[phase]] = True
                with open(os.path.join(self.out_path, 'success.lock'), 'a') as lockf:
                    fcntl.flock(lockf, fcntl.LOCK_EX)
                    try:
                        if os.path.exists(fn):
                            disk = np.load(fn)
                            if disk.shape == tab.shape:
                                tab |= disk.astype(bool)
                        tmp = fn[:-4] + uniq
                        np.save(tmp, tab)
                        os.replace(tmp, fn)
                    finally:
                        fcntl.flock(lockf, fcntl.LOCK_UN)

    def _stored(self, i, phase):
        return bool(self.checkpoint and self._success()[i, self._SUCCESS_COLS[phase]])

    def _load(self, i, key):
        return np.load(self._store_fn(i, key))

    # ------------------------------------------------------------------ phases
    def _device_of(self, i):
        if self.devices is None:
            from . import _ffi
            self.devices = list(range(_ffi.device_count()))
        return self.devices[i % len(self.devices)]

    def _spacing(self, i):
        meta = self._tile_meta[i]
        n = meta['shape'][0]
        if DEBUG:
            return dict(dX=np.ones(n - 1), dY=np.ones(n - 1), dX2=np.ones(n), dY2=np.ones(n))
        if 'dX' in meta:
            return dict(dX=np.asarray(meta['dX'], float), dY=np.asarray(meta['dY'], float),
                        dX2=np.asarray(meta.get('dX2', np.ones(n)), float), dY2=np.asarray(meta.get('dY2', np.ones(n)), float))
        # projected tiles: pixel size is the spacing (reference utils.mk_dx_dy_from_geotif_layer :132-137)
        a, e = meta['dlon'], meta['dlat']
        return dict(dX=np.ones(n - 1) * a, dY=np.abs(np.ones(n - 1) * e), dX2=np.ones(n) * a, dY2=np.abs(np.ones(n) * e))

    def _owned(self):
        return [i for i in range(self.n_inputs) if self.transport.owns(i)]

    def process_elevation(self, indices=None):
        """Reference :993-1008 + worker calc_elev_cond :54-71 (fill flats, drain pit paths, store elev)."""
        for i in self._owned():
            kw = dict(self.dem_proc_kwargs)
            kw.update(self._spacing(i))
            if self.elev_conditioned:
                kw['fill_flats'] = False
                kw['drain_pits_path'] = False
            meta = self._tile_meta[i]
            if self._stored(i, 'elev'):
                dp = self._make_processor(i, elev=self._load(i, 'elev'), **kw)        # the conditioned surface of an earlier run
            elif meta.get('synth') is not None:
                dp = self.processor_cls.from_synthetic(meta['shape'], meta['synth'], device=self._device_of(i), **kw)
            else:
                dp = self._make_processor(i, elev=meta['elev'], **kw)
            if not self.elev_conditioned and not self._stored(i, 'elev'):
                dp.calc_fill_flats()
                getattr(dp, 'run_pit_drain_paths', dp.calc_pit_drain_paths)()     # (the device processor's variant leaves the surface in HBM)
            self.tiles[i] = dp
            if not self._stored(i, 'elev') and self.checkpoint:
                self._store(i, 'elev', {'elev': np.asarray(dp.elev, float)})
        return [1] * self.n_inputs

    def _make_processor(self, i, **kw):
        if self.processor_cls is DEMProcessor:
            kw['device'] = self._device_of(i)
        return self.processor_cls(**kw)

    def _in_flight(self):
        """Worker threads for the per-tile phases: the configured number; by default one per GPU that holds tiles of this process
        and, since round 6, TWO per GPU that holds at least two of them: the stages of a tile are bound by different things (the pit
        search by instruction issue, the sweep by dependent latencies and traffic), so two tiles whose pipelines drift apart share
        the chip better than one -- measured on eight resident 16384^2 tiles: 395 -> 356 ms for the per-tile phases (two tiles:
        98.8 -> 91.5 ms, profiles/r06_two_tiles_in_flight_16384.txt).  Only with the device processor (a custom processor class
        need not be thread-safe)."""
        if self.tiles_in_flight is not None:
            return max(1, int(self.tiles_in_flight))
        per_dev = {}
        for i in self._owned():
            dev = getattr(self.tiles[i], '_device', None) if self.tiles[i] is not None else None
            if dev is not None:
                per_dev[dev] = per_dev.get(dev, 0) + 1
        if not per_dev:
            return 1
        if self.processor_cls is DEMProcessor:
            return sum(2 if n >= 2 else 1 for n in per_dev.values())
        return len(per_dev)

    def _per_tile(self, fn):
        """Run fn(i) for every owned tile: one after the other, or `tiles_in_flight` at a time from worker threads
        (the reference's n_workers pool, :1214-1288, with threads instead of processes: ctypes releases the GIL for
        the duration of a library call, and the latency-bound tail of one tile's sweep overlaps the next tile's
        streaming kernels on the same GPU)."""
        owned = list(self._owned())
        k = self._in_flight()
        if k == 1 or len(owned) < 2:
            for i in owned:
                fn(i)
            return
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=k) as ex:
            list(ex.map(fn, owned))                    # (re-raises a worker's exception)

    def process_aspect_slope(self):
        """Reference :1010-1030 + worker calc_aspect_slope :73-92."""
        self.compute_grid_overlaps()

        def one(i):
            dp = self.tiles[i]
            # "assuming we already did this" (:78) -- but the worker applies the caller's dem_proc_kwargs AFTER that line
            # (kwargs.update(dp_kwargs), :79), so an explicit fill_flats=True conditions the stored surface once more
            dp.fill_flats = bool(self.dem_proc_kwargs.get('fill_flats', False))
            if self._stored(i, 'aspect_slope'):
                dp.direction = self._load(i, 'aspect')
                dp.mag = self._load(i, 'slope')
                return
            getattr(dp, 'run_slopes_directions', dp.calc_slopes_directions)()
            if self.checkpoint:
                self._store(i, 'aspect_slope', {'aspect': dp.direction, 'slope': dp.mag})
        self._per_tile(one)
        return [1] * self.n_inputs

    def _patch_overlap1_edges(self):
        """Single-pixel-overlap fix of the reference's calc_uca worker (:102-180): an edge cell whose
        flow leaves the tile takes aspect/slope from the coincident (or adjacent) neighbour line.  The
        reference does this tile by tile in index order, writing the patched lines back to the shared
        store; the same order is kept here, on the edge lines only."""
        two_pi = 2 * np.pi
        downstream = {'left': [np.pi / 2, 3 * np.pi / 2], 'right': [two_pi - np.pi / 2, np.pi / 2],
                      'top': [0, np.pi], 'bottom': [np.pi, two_pi]}
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}       # (axis, index)
        for i in range(self.n_inputs):
            out_slice = self.grid_slice[i]
            for key in SIDES:
                if not self.check_1overlap(out_slice, self.edge_data[i][key]):
                    continue
                axis, idx = own[key]
                src, s_axis, s_idx = self._edge_line(i, key)
                nb_dir, nb_mag, d, mg = self._lines([(src, 'direction', s_axis, s_idx), (src, 'mag', s_axis, s_idx),
                                                     (i, 'direction', axis, idx), (i, 'mag', axis, idx)])
                if not self.transport.owns(i):
                    continue
                if key == 'right':
                    ids = ((d >= downstream[key][0]) | (d <= downstream[key][1])) & (d >= 0)
                else:
                    ids = (d >= downstream[key][0]) & (d <= downstream[key][1])
                if key in ('top', 'bottom'):
                    ids = ids | ((d < 1e-6) & (d >= 0)) | (np.abs(d - np.pi * 2) < 1e-6)
                ids = ids | (d == -1)
                if key in ('left', 'right'):
                    ids[0] = ids[0] & (not ((d[0] >= downstream['top'][0]) & (d[0] <= downstream['top'][1])))
                    ids[-1] = ids[-1] & (not ((d[-1] >= downstream['bottom'][0]) & (d[-1] <= downstream['bottom'][1])))
                else:
                    ids[0] = ids[0] & (not ((d[0] >= downstream['left'][0]) & (d[0] <= downstream['left'][1])))
                    ids[-1] = ids[-1] & (not ((d[-1] >= downstream['right'][0]) | (d[-1] <= downstream['right'][1])) & (d[-1] >= 0))
                if nb_dir is None:
                    nb_dir = np.zeros_like(d); nb_mag = np.zeros_like(d)
                d = d.copy(); mg = mg.copy()
                d[ids] = nb_dir[ids]
                mg[ids] = nb_mag[ids]
                self.tiles[i].set_line('direction', axis, idx, d)
                self.tiles[i].set_line('mag', axis, idx, mg)
            for key in CORNERS:
                if not self.check_1overlap(out_slice, self.edge_data[i][key]):
                    continue
                keytb, keylr = key.split('-')
                rr, cc = EDGE_SLICES[key]
                src, s_r, s_c = self._edge_line(i, key)
                (row_d, row_m, nb_d_row, nb_m_row) = self._lines([(i, 'direction', 0, rr), (i, 'mag', 0, rr),
                                                                  (src, 'direction', 0, s_r), (src, 'mag', 0, s_r)])
                if not self.transport.owns(i):
                    continue
                d = row_d[cc]
                tb_ok = ((d >= downstream[keytb][0]) & (d <= downstream[keytb][1])) | (((d < 1e-6) & (d >= 0)) | (np.abs(d - np.pi * 2) < 1e-6))
                if keylr == 'right':
                    ids = (((d >= downstream[keylr][0]) | (d <= downstream[keylr][1])) & (d >= 0)) & tb_ok
                else:
                    ids = (d >= downstream[keylr][0]) & (d <= downstream[keylr][1]) & tb_ok
                if not ids:
                    continue
                row_d = row_d.copy(); row_m = row_m.copy()
                row_d[cc] = 0.0 if nb_d_row is None else nb_d_row[s_c]
                row_m[cc] = 0.0 if nb_m_row is None else nb_m_row[s_c]
                self.tiles[i].set_line('direction', 0, rr, row_d)
                self.tiles[i].set_line('mag', 0, rr, row_m)

    def _lines(self, requests):
        return self.transport.gather_lines(requests)

    # During the edge fix-up a tile's lines (uca, edge_done, edge_todo) only change when that tile runs an
    # edge round, so they are fetched once and kept on the host; metrics refreshes then cost no device or
    # network round trip at all.
    def _edge_lines(self, requests):
        missing = [r for r in requests if r not in self._edge_cache]
        if missing:
            # collective transports must be entered by every rank with the same request list
            fetch = sorted(set(missing))
            for r, v in zip(fetch, self._lines(fetch)):
                self._edge_cache[r] = v
        return [self._edge_cache[r] for r in requests]

    def _edge_cache_drop(self, tile):
        for key in [k for k in self._edge_cache if k[0] == tile]:
            del self._edge_cache[key]

    def process_uca(self):
        """Reference :1032-1059 + worker calc_uca :94-197 (overlap-1 patch, find_flats, calc_uca)."""
        self._patch_overlap1_edges()
        if self.processor_cls is DEMProcessor:
            from . import _ffi
            _ffi.release_scratch()         # the conditioning stages are over: their per-device arena (10-20 GB after a large tile) goes back

        def one(i):
            dp = self.tiles[i]
            dp.find_flats()
            self.uca0[i] = None
            if self._stored(i, 'uca'):
                # first-pass area and edge masks of an earlier run, plus whatever edge corrections it had reached
                if hasattr(dp, 'build_graph'):
                    dp.build_graph()
                    dp.restore_pit_slopes()
                first = self._load(i, 'uca')
                dp.uca = first + self._load(i, 'uca_edges') if os.path.exists(self._store_fn(i, 'uca_edges')) else first
                dp.edge_todo = self._load(i, 'edge_todo')
                dp.edge_done = self._load(i, 'edge_done')
                if self.keep_first_pass_uca:
                    self.uca0[i] = first
                return
            getattr(dp, 'run_uca', dp.calc_uca)()
            dp.restore_pit_slopes()        # the worker does not write its patched slope back (:192-194)
            if self.checkpoint:
                self._store(i, 'uca', {'uca': dp.uca, 'edge_todo': dp.edge_todo, 'edge_done': dp.edge_done})
        self._per_tile(one)
        return [1] * self.n_inputs

    # ---- edge fix-up ------------------------------------------------------------------------
    def _edge_inputs(self, i, snap, drop_mutual_todo=True):
        """Strips for one tile from the snapshot of the previous round, with the corner rules of the
        reference's calc_uca_ec (:250-274).  drop_mutual_todo=False leaves out the last rule (:274)."""
        data, done, todo, todo_nb = {}, {}, {}, {}
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        for key in SIDES:
            src, axis, idx = self._edge_line(i, key)
            L = self.tiles_shape[i][0] if axis == 1 else self.tiles_shape[i][1]
            if src < 0:
                data[key] = np.zeros(L); done[key] = np.zeros(L, bool); todo_nb[key] = np.zeros(L, bool)
            else:
                data[key] = snap[(src, 'uca', axis, idx)].copy()
                done[key] = snap[(src, 'edge_done', axis, idx)].copy()
                todo_nb[key] = snap[(src, 'edge_todo', axis, idx)].copy()
            a, ix = own[key]
            todo[key] = snap[(i, 'edge_todo', a, ix)].copy()
        for key in ('top-left', 'bottom-right', 'top-right', 'bottom-left'):            # order of the reference (:258)
            keytb, keylr = key.split('-')
            inds = EDGE_SLICES[key]
            overlap = done[keytb][inds[1]] & done[keylr][inds[0]]
            if overlap:
                done[keytb][inds[1]] = False                                             # :266
                if self.check_1overlap(self.grid_slice[i], self.edge_data[i][key]):
                    src, lr, lc = self._edge_line(i, key)
                    if src >= 0 and snap[(src, 'edge_done', 0, lr)][lc]:                 # :268
                        v = snap[(src, 'uca', 0, lr)][lc]
                        data[keylr][inds[0]] = v                                         # :269
                        data[keytb][inds[1]] = v                                         # :270
        if drop_mutual_todo == 'self':
            # only where the "neighbour" line is the tile's own edge (the mosaic border, :695-705 with no overlap found)
            todo = {k: (v & (todo_nb[k] == False)) if self._edge_line(i, k)[0] == i else v for k, v in todo.items()}   # noqa: E712
        elif drop_mutual_todo:
            todo = {k: v & (todo_nb[k] == False) for k, v in todo.items()}               # noqa: E712  (:274)
        return data, done, todo

    def _snapshot_requests(self, i):
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        req = set()
        for key in SIDES:
            src, axis, idx = self._edge_line(i, key)
            if src >= 0:
                for nm in ('uca', 'edge_done', 'edge_todo'):
                    req.add((src, nm, axis, idx))
            a, ix = own[key]
            req.add((i, 'edge_todo', a, ix))
            req.add((i, 'edge_done', a, ix))                                             # pool mode: `_adopt_finished`
        for key in CORNERS:
            if self.check_1overlap(self.grid_slice[i], self.edge_data[i][key]):
                src, lr, lc = self._edge_line(i, key)
                if src >= 0:
                    req.add((src, 'edge_done', 0, lr)); req.add((src, 'uca', 0, lr))
        return req

    def _metric_requests(self, i):
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        req = set()
        for key in SIDES:
            src, axis, idx = self._edge_line(i, key)
            if src >= 0:
                req.add((src, 'edge_done', axis, idx))
            req.add((i, 'edge_todo',) + own[key])
        for key in CORNERS:
            src, lr, lc = self._edge_line(i, key)
            if src >= 0:
                req.add((src, 'edge_done', 0, lr))
        return req

    def _tile_metric(self, i, snap):
        """(fraction, count) of tile i's 'todo' edge cells whose neighbour cell is done; the four corner
        pixels count too (reference calc_uca_ec_metrics :199-221 iterates all 8 edge_slice keys)."""
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        n_done = p_done = 0
        for key in SIDES:
            src, axis, idx = self._edge_line(i, key)
            et = snap[(i, 'edge_todo',) + own[key]]
            edn = snap[(src, 'edge_done', axis, idx)] if src >= 0 else np.zeros_like(et)
            n_done += int((et & edn).sum())
            p_done += int(et.sum())
        for key in CORNERS:
            keytb, keylr = key.split('-')
            rr, cc = EDGE_SLICES[key]
            et = bool(snap[(i, 'edge_todo',) + own[keytb]][cc])
            src, lr, lc = self._edge_line(i, key)
            edn = bool(snap[(src, 'edge_done', 0, lr)][lc]) if src >= 0 else False
            n_done += int(et and edn)
            p_done += int(et)
        return (n_done / (1e-16 + p_done), n_done)

    def update_uca_edge_metrics(self, index=None):
        """Refresh the per-tile edge metrics table (reference :1061-1088) for the tiles in `index`."""
        if index is None:
            index = range(self.n_inputs)
        if getattr(self, '_mets', None) is None or len(self._mets) != self.n_inputs:
            self._mets = np.zeros((self.n_inputs, 2))
        reqs = set()
        for i in index:
            reqs |= self._metric_requests(i)
        reqs = sorted(reqs)
        snap = dict(zip(reqs, self._edge_lines(reqs)))
        for i in index:
            self._mets[i] = self._tile_metric(i, snap)
        return self._mets.copy()

    def _edge_round(self, i):
        """One calc_uca_ec of the reference (:224-284) for tile i, from the neighbours' current lines."""
        reqs = sorted(self._snapshot_requests(i))
        snap = dict(zip(reqs, self._edge_lines(reqs)))
        data, done, todo = self._edge_inputs(i, snap)
        # An edge round is a pure function of (tile state, strips).  If this tile already ran a round
        # with exactly these strips and nothing changed since, the round would reproduce the same
        # state (finished edge cells re-synchronised to the same values, same masks): skip it.  The
        # reference's loop revisits tiles many times before its ranking settles (:1142-1211).
        sig = tuple(np.asarray(v[k]).tobytes() for v in (data, done, todo) for k in SIDES)
        if self._edge_last.get(i) == sig:
            self.edge_rounds_skipped += 1
            return
        self._edge_last[i] = sig
        self._edge_cache_drop(i)
        self._run_edge_round(i, data, done, todo)

    def _run_edge_round(self, i, data, done, todo, incremental=False):
        """dp.calc_uca(uca_init=uca + uca_edges, edge_init_data=...) of the worker (:276-279) on the resident tile.
        incremental (pool waves, device processor): the round runs on the fix-up state the tile keeps between rounds."""
        if not self.transport.owns(i):
            return
        dp = self.tiles[i]
        t0 = time.perf_counter()
        try:
            self._run_edge_round_inner(i, dp, data, done, todo, incremental)
        finally:
            self.edge_round_log.append((self.edge_waves, i, (time.perf_counter() - t0) * 1e3))

    def _run_edge_round_inner(self, i, dp, data, done, todo, incremental=False):
        if self.keep_first_pass_uca and self.uca0[i] is None:
            self.uca0[i] = np.array(dp.uca)              # the reference keeps the first pass as 'uca'
        if hasattr(dp, 'run_uca'):
            dp.run_uca(edge_init_data=[data, done, todo], uca_resident=True, incremental=incremental)
        else:
            dp.calc_uca(uca_init=dp.uca, edge_init_data=[data, done, todo])

    def _rank_tiles(self, mets, mets_type):
        if mets.shape[0] == 1:
            return np.zeros(1, int)
        return np.argpartition(-mets[:, mets_type], min(self.n_workers * 2, mets.shape[0] - 1))   # :1111-1113

    def _neighbours(self, f):
        """Tile f and its four side neighbours (the metrics the reference refreshes after a round, check_mets :1116-1136)."""
        nr, nc = self.grid_id2i.shape
        r, c, _ = self.grid_id[f]
        out = [int(f)]
        if c > 0: out.append(int(self.grid_id2i[r, c - 1]))
        if c < nc - 1: out.append(int(self.grid_id2i[r, c + 1]))
        if r > 0: out.append(int(self.grid_id2i[r - 1, c]))
        if r < nr - 1: out.append(int(self.grid_id2i[r + 1, c]))
        return [t for t in out if t != -1]

    def _edge_setup(self):
        self.tiles_shape = [tuple(int(v) for v in self.index[i, 6:]) for i in range(self.n_inputs)]
        self.edge_rounds = 0
        self.edge_rounds_skipped = 0
        self.edge_waves = 0
        self.edge_round_log = []           # (wave, tile, host ms) of every round this process ran
        self.edge_queued_batches = 0       # batches of waves chosen on the device and queued (pydem_board_run_waves)
        self._edge_line_memo = {}
        self._mets = None
        self._edge_cache = {}
        self._edge_last = {}
        self.transport_is_collective = not type(self.transport) is EdgeTransport
        # every line of a tile that some round or metric of another (or the same) tile can ask for: after a tile ran
        # its round they are refreshed in ONE batch (one device synchronisation / one collective per round or wave)
        interest = {}
        for i in range(self.n_inputs):
            for req in self._snapshot_requests(i) | self._metric_requests(i):
                interest.setdefault(req[0], set()).add(req)
        return interest

    def process_uca_edges(self, mets_type=0):
        """Cross-tile UCA correction (reference :1090-1246).  Two schedules, like the reference:

        edge_mode='reference' (default when n_workers == 1) follows the reference's serial loop exactly
        (:1140-1211): rank the tiles by the fraction of their 'todo' edge cells that face a finished
        neighbour, run one edge round on the first (even when its count is zero), refresh the metrics of
        that tile and its four neighbours, stop when the ranking no longer changes.  The result depends on
        the visiting order (finished edge cells are re-synchronised with the neighbour's value at every
        round, :806-809, and :274 drops a tile's 'todo' where the neighbour is 'todo' too), so the pm_*
        goldens of the reference are reproduced in this mode only.

        edge_mode='pool' (default when n_workers > 1): `_process_uca_edges_pool`."""
        mode = getattr(self, 'edge_mode', None) or ('reference' if self.n_workers == 1 else 'pool')
        if mode == 'pool':
            return self._process_uca_edges_pool(mets_type)
        if mode != 'reference':
            raise ValueError("edge_mode must be 'reference' or 'pool', not %r" % (mode,))
        interest = self._edge_setup()
        mets = self.update_uca_edge_metrics()
        I = self._rank_tiles(mets, mets_type)
        I_old = np.zeros_like(I)
        while np.any(I_old != I) and self.edge_rounds < self.max_edge_rounds:
            f = int(I[0])
            self._edge_round(f)
            self._edge_lines(sorted(interest.get(f, ())))
            self.edge_rounds += 1
            self.edge_waves += 1
            I_old[:] = I[:]
            mets = self.update_uca_edge_metrics(sorted(set(self._neighbours(f))))
            I = self._rank_tiles(mets, mets_type)
        return mets

    def _todo_dropped(self, i, snap, todo):
        """Number of tile i's 'todo' edge PIXELS that a round with the per-side flags `todo` (after rule :274, from
        `_edge_inputs`) would drop: a pixel survives when any side it belongs to keeps it (corner pixels sit on two
        strips, :726-739 ORs them)."""
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        kept = {k: todo[k].copy() for k in SIDES}
        for keytb, keylr in (('top', 'left'), ('top', 'right'), ('bottom', 'left'), ('bottom', 'right')):
            a = 0 if keytb == 'top' else -1          # position of the corner on the left/right strips
            b = 0 if keylr == 'left' else -1         # ... and on the top/bottom strips
            kept[keylr][a] = kept[keytb][b] = kept[keylr][a] | kept[keytb][b]
        lost = {k: snap[(i, 'edge_todo',) + own[k]] & ~kept[k] for k in SIDES}
        n = sum(int(np.count_nonzero(v)) for v in lost.values())
        for keytb, keylr in (('top', 'left'), ('top', 'right'), ('bottom', 'left'), ('bottom', 'right')):
            a = 0 if keytb == 'top' else -1
            b = 0 if keylr == 'left' else -1
            n -= int(lost[keylr][a] and lost[keytb][b])                             # a lost corner was counted twice
        return n

    def _adopt_finished(self, i, snap, done, todo):
        """Pool mode: an edge cell whose neighbour cell is finished while the tile's own value is not (edge_done False:
        it lies downstream of one of the tile's unresolved inlets without being an inlet itself) becomes a seed --
        it adopts the neighbour's final value and hands the difference downstream like any other seed.  The worker of
        the reference only re-synchronises such a cell with the neighbour's value (:806-809) AND lets it receive from
        the seeds upstream of it in the same round (it is not 'done on the tile edge', cyutils.pyx:159-161): the
        upstream area is then counted twice.  Whether the serial loop runs into this depends on its visiting order
        (a corner pixel shared by four one-pixel-overlap tiles of the reference's cone test does, for some orders)."""
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        return {k: todo[k] | (done[k] & ~snap[(i, 'edge_done',) + own[k]]) for k in SIDES}

    def _process_uca_edges_pool(self, mets_type=0):
        """The reference's multi-worker schedule (:1214-1246) as deterministic waves.  The reference keeps up to
        2 * n_workers edge rounds in flight: the best-ranked tiles (np.argpartition of the metric, :1111-1113) whose
        count of 'todo' cells facing a finished neighbour is positive; every finished worker refreshes the metrics of
        its tile and its neighbours (check_mets) and the free slots are refilled from the new ranking; the loop ends
        when nothing is in flight and no candidate is left.  Which strips a worker reads depends on when the OS
        starts it, so the reference's results are not reproducible run to run.  Here a WAVE takes the same candidates,
        builds all their strip inputs from ONE snapshot, runs the rounds concurrently (tiles of different GPUs / ranks
        at the same time), then refreshes the lines and metrics of everything that ran -- one strip exchange per wave.
        Finished cells are final, so tiles that update side by side from one snapshot only see each other's values one
        wave late.  What is NOT safe to do early is rule :274 of the worker: it drops a tile's 'todo' flag where the
        neighbour's line is 'todo' too ("that should never happen except for floating point rounding errors") -- but
        it also happens, legitimately, wherever the neighbour's cell sits on that neighbour's own unresolved inlet
        edge (line ends, one-pixel overlaps) until the neighbour has run, and a dropped flag turns the cell and
        everything downstream of it into 'done' with the upstream area missing.  The serial loop gets away with it by
        visiting the tiles with the highest fraction first (when it does; on terrain with pits it loses cells, and
        which ones depends on the visiting order).  The waves therefore apply :274 only as the tie-break it was
        written for: 'todo' cells on the mosaic border (where the edge table points a tile at its own line) are dropped
        at once -- a tile that has such cells is a candidate even with a zero count, the serial loop only cleans the
        tiles it happens to visit -- while any tile has a seed, other mutual 'todo' cells stay 'todo'; when no tile can make progress
        and :274 would still drop cells somewhere, ONE tile (the one that would drop most, lowest index first) runs a round with
        :274 and gives its cells up, which seeds its neighbours.  A tile whose inputs are bit-identical to those of
        its last round is not a candidate (the round would reproduce its state; the reference would re-run it for ever
        when only a corner pixel keeps its count > 0)."""
        if self._device_board_usable():
            return self._process_uca_edges_pool_device(mets_type)
        interest = self._edge_setup()
        mets = self.update_uca_edge_metrics()
        width = max(1, 2 * int(self.n_workers))
        self.edge_tiebreaks = 0
        inc = bool(getattr(self, 'edge_incremental', True))
        while self.edge_waves < self.max_edge_rounds:
            inputs, snaps = {}, {}
            eff = np.zeros_like(mets)
            for a in range(self.n_inputs):
                reqs = sorted(self._snapshot_requests(a))
                snaps[a] = dict(zip(reqs, self._edge_lines(reqs)))
                data, done, todo = self._edge_inputs(a, snaps[a], drop_mutual_todo='self')
                if mets[a, 0] <= 0 and self._todo_dropped(a, snaps[a], todo) == 0:
                    continue
                todo = self._adopt_finished(a, snaps[a], done, todo)
                # (values of unfinished neighbour cells never enter a round: they are not part of the signature)
                sig = tuple(np.where(done[k], data[k], 0.0).tobytes() for k in SIDES) + \
                    tuple(np.asarray(v[k]).tobytes() for v in (done, todo) for k in SIDES)
                if self._edge_last.get(a) == sig:
                    continue
                inputs[a] = (data, done, todo, sig)
                eff[a] = mets[a] if mets[a, 0] > 0 else (1e-9, 0)       # border clean-up rounds rank last
            if inputs:
                I = self._rank_tiles(eff, mets_type)
                wave = [int(a) for a in I[:width] if a in inputs]
                wave.sort()
            else:
                # nobody can make progress: the tie-break of :274, one tile at a time
                best = None
                for a in range(self.n_inputs):
                    data, done, todo = self._edge_inputs(a, snaps[a], drop_mutual_todo=True)
                    n = self._todo_dropped(a, snaps[a], todo)
                    if n > 0 and (best is None or n > best[0]):
                        best = (n, a, data, done, self._adopt_finished(a, snaps[a], done, todo))
                if best is None:
                    break
                a = best[1]
                inputs[a] = best[2:] + (None,)
                wave = [a]
                self.edge_tiebreaks += 1
            for a in wave:
                self._edge_last[a] = inputs[a][3]
            mine = [a for a in wave if self.transport.owns(a)]
            k = self._in_flight()
            if k > 1 and len(mine) > 1:
                # one process driving several GPUs (or several tiles of one GPU): the rounds of a wave from worker threads
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=k) as ex:
                    list(ex.map(lambda a: self._run_edge_round(a, *inputs[a][:3], incremental=inc), mine))
            else:
                for a in mine:
                    self._run_edge_round(a, *inputs[a][:3], incremental=inc)
            fetch = set()
            for a in wave:
                self._edge_cache_drop(a)
                fetch |= interest.get(a, set())
            self._edge_lines(sorted(fetch))
            self.edge_rounds += len(wave)
            self.edge_waves += 1
            check = set()
            for a in wave:
                check.update(self._neighbours(a))
            mets = self.update_uca_edge_metrics(sorted(check))
        for a in self._owned():                   # incremental rounds: cells still below an unresolved inlet catch up
            if hasattr(self.tiles[a], 'flush_edge_rounds'):
                self.tiles[a].flush_edge_rounds()
        return mets

    # ---- the same schedule with the strips resident on the device ------------------------------------------
    def _board_device(self):
        """The GPU this process keeps its replica of the edge board on: the one its tiles live on; a rank that owns no
        tile (more ranks than tiles) still takes part in the board's collectives, on its communicator's device."""
        owned = self._owned()
        if owned:
            return self.tiles[owned[0]]._device
        comm = getattr(self.transport, 'comm', None)
        return getattr(comm, 'device', self._device_of(getattr(self.transport, 'rank', 0)))

    def _device_board_usable(self):
        """Device processors, all of this process's tiles on one GPU, strips carried in-process or by RCCL.  The
        answer is COLLECTIVE: the host pool path and the device board issue different collectives, so every rank must
        take the same branch -- each rank judges its own tiles (a rank without tiles has no objection) and the ranks
        agree on the minimum."""
        ok = bool(getattr(self, 'edge_device_board', True) and getattr(self, 'edge_incremental', True))
        owned = self._owned()
        ok = ok and all(hasattr(self.tiles[i], 'run_edge_round_dev') for i in owned)
        ok = ok and not self.dem_proc_kwargs.get('apply_uca_limit_edges')
        ok = ok and len(set(self.tiles[i]._device for i in owned)) <= 1
        if type(self.transport) is EdgeTransport:
            return ok and len(owned) == self.n_inputs and self.n_inputs > 0
        if not (hasattr(getattr(self.transport, 'comm', None), '_h') or hasattr(self.transport, 'sum_inplace')):
            ok = False                                         # neither RCCL nor a host sum for the staging buffer
        return self.transport.allreduce_max(0.0 if ok else 1.0) == 0.0

    def _process_uca_edges_pool_device(self, mets_type=0):
        """`_process_uca_edges_pool` with the edge board of csrc/comm.hip: the lines every tile reads live in one
        replicated device buffer, a kernel applies the strip rules above (they stay the specification; the CPU tier
        and tests/test_gpu_process_manager.py hold the two against each other) and the host only sees a few numbers
        per tile and wave.  Same candidates, same tie-break, same waves."""
        from . import _ffi
        from .dem_processing import _FIELD_OF
        interest = self._edge_setup()
        n_t = self.n_inputs
        own = {'left': (1, 0), 'right': (1, -1), 'top': (0, 0), 'bottom': (0, -1)}
        length = lambda req: int(self.index[req[0], 7] if req[2] == 0 else self.index[req[0], 6])
        layout, start, size, off = {}, {}, {}, 0
        for t in range(n_t):
            start[t] = off
            for req in sorted(interest.get(t, ())):
                layout[req] = off
                off += length(req)
            size[t] = off - start[t]
        owned = self._owned()
        comm = getattr(self.transport, 'comm', None)
        board = _ffi.Board(self._board_device(), n_t, off)
        readers = {t: set([t]) for t in range(n_t)}
        for i in range(n_t):
            o28, f8 = [], []
            srcs = [self._edge_line(i, key) for key in SIDES]
            o28 += [layout[(i, 'edge_todo',) + own[key]] for key in SIDES]
            o28 += [layout[(i, 'edge_done',) + own[key]] for key in SIDES]
            for name in ('uca', 'edge_done', 'edge_todo'):
                o28 += [layout[(src, name, axis, idx)] if src >= 0 else -1 for src, axis, idx in srcs]
            f8 += [int(src == i) for src, _, _ in srcs]
            cd, cu, c1 = [], [], []
            for key in ('top-left', 'top-right', 'bottom-left', 'bottom-right'):
                src, lr, lc = self._edge_line(i, key)
                ov1 = bool(self.check_1overlap(self.grid_slice[i], self.edge_data[i][key]))
                cd.append(layout[(src, 'edge_done', 0, lr)] + lc if src >= 0 else -1)
                cu.append(layout[(src, 'uca', 0, lr)] + lc if (src >= 0 and ov1) else -1)
                c1.append(int(ov1))
                if src >= 0:
                    readers[src].add(i)
            for src, _, _ in srcs:
                if src >= 0:
                    readers[src].add(i)
            n, m = self.tiles_shape[i]
            dp = self.tiles[i] if self.transport.owns(i) else None
            if dp is not None:
                dp._ensure_tile()
                dp._push('elev', 'mag', 'direction', 'uca', 'edge_todo', 'edge_done', 'flats')
            board.set_desc(i, n, m, o28 + cd + cu, f8 + c1, dp._tile if dp is not None else None)

        for t in range(n_t):
            if self.transport.owns(t):
                board.set_lines(t, start[t], size[t], self.tiles[t]._tile,
                                [(_FIELD_OF[req[1]], req[2], req[3], layout[req] - start[t]) for req in sorted(interest.get(t, ()))])
            else:                          # (a tile of another rank: only the layout of its lines -- which are areas, which masks)
                board.set_lines(t, start[t], size[t], None,
                                [(_FIELD_OF[req[1]], req[2], req[3], layout[req] - start[t]) for req in sorted(interest.get(t, ()))])

        host_sum = None if (comm is not None or type(self.transport) is EdgeTransport) else self.transport.sum_inplace

        def refresh(tiles):
            board.refresh(comm, sorted(tiles), host_sum)

        def check_against_host_rules(tiles, scal):
            # PYDEM_BOARD_CHECK=1 (tests): the numbers of the evaluation kernel against the host rules on the same lines
            self._edge_cache = {}
            for a in tiles:
                reqs = sorted(self._snapshot_requests(a) | self._metric_requests(a))
                snap = dict(zip(reqs, self._edge_lines(reqs)))
                met = self._tile_metric(a, snap)
                got = [int(v) for v in scal[a, :5]]
                want = [met[1], None]
                d0, dn0, td0 = self._edge_inputs(a, snap, drop_mutual_todo='self')
                d1, dn1, td1 = self._edge_inputs(a, snap, drop_mutual_todo=True)
                want += [self._todo_dropped(a, snap, td0), self._todo_dropped(a, snap, td1)]
                tda = self._adopt_finished(a, snap, dn0, td0)
                want.append(sum(int(np.count_nonzero(dn0[k] & tda[k])) for k in SIDES))
                for j in (0, 2, 3, 4):
                    assert got[j] == want[j], "edge board: tile %d value %d: kernel %r, host rules %r" % (a, j, got, want)
                assert abs(got[0] / (1e-16 + got[1]) - met[0]) < 1e-12, (a, got, met)
            self._edge_cache = {}

        checking = bool(os.environ.get('PYDEM_BOARD_CHECK'))
        refresh(range(n_t))
        scal = board.eval(list(range(n_t)), [0] * n_t)
        if checking:
            check_against_host_rules(range(n_t), scal)
        width = max(1, 2 * int(self.n_workers))
        self.edge_tiebreaks = 0
        last_hash = {}
        built = set()                      # tiles whose first round (the one that builds the fix-up state) has run
        mets = np.zeros((n_t, 2))
        def refresh_mets(tiles):
            for a in tiles:
                nd = float(scal[a, 0])
                mets[a] = (nd / (1e-16 + float(scal[a, 1])), nd)

        refresh_mets(range(n_t))
        prof = {'select': 0.0, 'rounds': 0.0, 'refresh': 0.0, 'eval': 0.0, 'queued': 0.0} if os.environ.get('PYDEM_EDGE_PROFILE') else None
        tp = time.perf_counter()

        def lap(key):                      # PYDEM_EDGE_PROFILE=1: host wall-clock of the wave loop by part
            nonlocal tp
            if prof is not None:
                now = time.perf_counter(); prof[key] += now - tp; tp = now
        # ---- queued waves (pydem_board_run_waves): with at most `width` tiles the ranking selects every candidate, so a kernel
        # can choose the wave and K waves run back to back behind one another with ONE look from the host per batch -- the
        # manager's poll / re-rank loop (:1214-1246) is off the critical path; the tie-break wave (no candidate left: the tile
        # that drops the most 'todo' pixels under rule :274 runs alone) is chosen by the same kernel.  The host still runs a
        # tile's first round (it builds the tile's fix-up state) and rounds that are not in the condensed form.  Same waves,
        # same rounds (tests/test_gpu_process_manager.py holds both loops against each other).
        k_queue = int(os.environ.get('PYDEM_EDGE_QUEUE', '16'))
        # (a transport without RCCL that can sum bytes -- the socket group -- drives the SAME queued path with the staging buffer
        # summed on the host once per wave: slower than its host-driven loop, but it is how the per-rank wave selection is tested
        # with more than one rank on one GPU; PYDEM_EDGE_QUEUE_HOST=0 keeps such transports on the host-driven loop)
        exchange = None
        if host_sum is not None and hasattr(self.transport, 'sum_bytes_inplace') and os.environ.get('PYDEM_EDGE_QUEUE_HOST', '1') != '0':
            exchange = self.transport
        # (`keep_first_pass_uca` does not stand in the way: a tile's FIRST round is always run by the host loop below -- it builds
        # the tile's fix-up state -- and that is where the first-pass area is put aside)
        queue = (k_queue > 0 and n_t <= min(width, 64) and not checking
                 and (host_sum is None or exchange is not None)
                 and all(hasattr(self.tiles[i]._tile, 'edge_queue_ready') for i in owned))
        staged = comm is not None or exchange is not None
        k_queue = min(k_queue, 64)
        ran = set()                        # tiles that have run a round (on any rank: the waves are the same everywhere)
        ran_ok = {}                        # len(ran) -> every rank can queue the rounds of its tiles in `ran`
        host_next = False                  # the last batch stopped in front of a wave only the host can run
        self.edge_wave_graphs = False      # the queued waves ran as captured hipGraphs
        self.edge_host_looks = 1           # times the host waited for the device inside the wave loop (+ the first evaluation)
        S = _ffi.Board
        sched = np.zeros(S.SCH_WORDS, np.uint64)
        if queue:
            for a in range(n_t):
                sched[S.SCH_READERS + a] = sum(1 << int(r) for r in readers[a])
                sched[S.SCH_NBRS + a] = sum(1 << int(r) for r in set(self._neighbours(a)))

        queued_batches = [0]

        def queued_batch():
            """Run up to k_queue waves on the device; True if the host has to run the next wave itself."""
            nonlocal scal
            queued_batches[0] += 1
            sched[S.SCH_OK] = sum(1 << a for a in ran)
            sched[S.SCH_LIMIT] = max(0, int(self.max_edge_rounds) - int(self.edge_waves))
            for a in range(n_t):
                sched[S.SCH_ND + a] = int(mets[a, 1])
                sched[S.SCH_HAS + a] = int(a in last_hash)
                sched[S.SCH_HASH + a] = last_hash.get(a, 0)
            for a in range(n_t):           # (the denominator the host's metric was formed with; only carried, never compared)
                sched[S.SCH_PD + a] = pd_host[a]
            # (while tiles are still missing their first round a batch ends early -- stop 2 -- and, with a communicator, its
            # remaining waves are no-op collectives of the whole staging buffer: short batches until every tile has run)
            kq = k_queue if len(ran) >= n_t else min(k_queue, 4)
            scal = board.run_waves(comm, kq, sched, exchange=exchange)
            self.edge_host_looks += 1 if exchange is None else int(sched[S.SCH_NWAVES]) + 1
            self.edge_wave_graphs = bool(sched[S.SCH_GRAPH])
            for w in range(int(sched[S.SCH_NWAVES])):
                members = [a for a in range(n_t) if (int(sched[S.SCH_LOG + w]) >> a) & 1]
                for a in members:
                    if self.transport.owns(a):
                        self.edge_round_log.append((self.edge_waves, a, 0.0))
                self.edge_rounds += len(members)
                self.edge_waves += 1
            self.edge_tiebreaks += int(sched[S.SCH_NTB])
            for a in range(n_t):
                nd, pd = int(sched[S.SCH_ND + a]), int(sched[S.SCH_PD + a])
                pd_host[a] = pd
                mets[a] = (float(nd) / (1e-16 + float(pd)), float(nd))
                if int(sched[S.SCH_HAS + a]):
                    last_hash[a] = int(sched[S.SCH_HASH + a])
                else:
                    last_hash.pop(a, None)            # (a tie-break wave forgot the tile's last strips)
            return int(sched[S.SCH_STOP]) in (1, 2)

        pd_host = [int(scal[a, 1]) for a in range(n_t)]
        while self.edge_waves < self.max_edge_rounds:
            if queue and ran and not host_next:
                if len(ran) not in ran_ok:
                    # everything that can fail on one rank alone happens HERE, and the ranks agree on the outcome, before any of
                    # them enters a batch (its collectives would wait for ever for a rank that backed out)
                    mine_ok = all(self.tiles[a]._tile.edge_queue_ready() for a in ran if self.transport.owns(a))
                    if mine_ok:
                        try:
                            board.prepare_waves(staged, sum(1 << a for a in ran))
                        except _ffi.HipError as exc:
                            logger.warning("queued edge waves: %s -- host-driven waves instead", exc)
                            mine_ok = False
                    ran_ok[len(ran)] = self.transport.allreduce_max(0.0 if mine_ok else 1.0) == 0.0
                if ran_ok[len(ran)]:
                    host_next = queued_batch()
                    lap('queued')
                    continue
            host_next = False
            eff = np.zeros_like(mets)
            cand = []
            for a in range(n_t):
                if mets[a, 0] <= 0 and scal[a, 2] == 0:
                    continue
                if last_hash.get(a) == int(scal[a, 5]):
                    continue
                cand.append(a)
                eff[a] = mets[a] if mets[a, 0] > 0 else (1e-9, 0)
            if cand:
                I = self._rank_tiles(eff, mets_type)
                cs = set(cand)
                wave = sorted(int(a) for a in I[:width] if a in cs)
                for a in wave:
                    last_hash[a] = int(scal[a, 5])
            else:
                drops = scal[:, 3].astype(np.int64)
                if not (drops > 0).any():
                    break
                a = int(np.argmax(drops))                   # the most dropped pixels, lowest index first
                board.eval([a], [1])                        # its strips again, with rule :274 everywhere
                last_hash.pop(a, None)
                wave = [a]
                self.edge_tiebreaks += 1
            mine = [a for a in wave if self.transport.owns(a)]
            for a in mine:
                dp = self.tiles[a]
                if self.keep_first_pass_uca and self.uca0[a] is None:
                    self.uca0[a] = np.array(dp.uca)
            k = self._in_flight()
            lap('select')

            def one(a):
                t0 = time.perf_counter()
                self.tiles[a].run_edge_round_dev()
                self.edge_round_log.append((self.edge_waves, a, (time.perf_counter() - t0) * 1e3))
            # a tile's FIRST round builds its fix-up state on the host (compact records + the condensed graph of the watched
            # cells, csrc/uca_cond.inl: ~25 ms of C++ per 16384^2 tile, the GIL is released): tiles of one process build side by side
            fresh = [a for a in mine if a not in built]
            built.update(mine)
            # (only when the user has not bounded the concurrency -- `tiles_in_flight` -- and the processor is ours: a custom
            # processor class or transport need not be thread-safe, and every build allocates pinned staging)
            widen = self.tiles_in_flight is None and self.processor_cls is DEMProcessor
            kk = max(k, min(len(fresh), 8)) if (widen and len(fresh) > 1) else k
            if kk > 1 and len(mine) > 1:
                from concurrent.futures import ThreadPoolExecutor
                with ThreadPoolExecutor(max_workers=kk) as ex:
                    list(ex.map(one, mine))
            else:
                for a in mine:
                    one(a)
            lap('rounds')
            refresh(wave)
            lap('refresh')
            affected = set()
            for a in wave:
                affected |= readers[a]
            affected = sorted(affected)
            scal = board.eval(affected, [0] * len(affected))
            lap('eval')
            if checking:
                check_against_host_rules(range(n_t), scal)
            self.edge_rounds += len(wave)
            self.edge_waves += 1
            check = set()
            for a in wave:                      # like check_mets (:1116-1136): the tiles that ran and their four side
                check.update(self._neighbours(a))   # neighbours; a diagonal neighbour keeps its old metric until then
            refresh_mets(sorted(check))
            for a in sorted(check):
                pd_host[a] = int(scal[a, 1])
            ran.update(wave)
            self.edge_host_looks += 1
        kf = min(len(owned), 8) if (self.tiles_in_flight is None and self.processor_cls is DEMProcessor) else min(len(owned), self._in_flight())
        if kf > 1:                         # the interiors catch up: one latency-bound cascade per tile, side by side on their streams
            from concurrent.futures import ThreadPoolExecutor
            with ThreadPoolExecutor(max_workers=kf) as ex:
                list(ex.map(lambda a: self.tiles[a].flush_edge_rounds(), owned))
        else:
            for a in owned:
                self.tiles[a].flush_edge_rounds()
        if prof is not None:
            import sys
            for a in owned:
                self.tiles[a]._tile.synchronize()
            prof['flush'] = time.perf_counter() - tp
            self.edge_profile = dict((k2, v * 1e3) for k2, v in prof.items())
            sys.stderr.write("edge fix-up wave loop (host ms): %s over %d waves, %d host looks\n"
                             % (', '.join('%s %.1f' % (k2, v * 1e3) for k2, v in prof.items()), self.edge_waves, self.edge_host_looks))
        self._mets = mets.copy()
        self.edge_queued_batches = queued_batches[0]
        # what every rank must agree on when the fix-up is over: the waves as they ran (members wave by wave: the log of this
        # rank's rounds + the counts) and -- `edge_board_digest = True`, bench.py's warm-up -- the replicated board itself
        self.edge_schedule_digest = (int(self.edge_waves), int(self.edge_rounds), int(self.edge_tiebreaks))
        if getattr(self, 'edge_board_digest', False):
            import hashlib
            self.edge_board_sha256 = hashlib.sha256(board.download().tobytes()).hexdigest()
        board.close()
        return mets

    def process_twi(self):
        """Full directory flow (reference :1290-1317)."""
        logger.info("Compute Grid")
        self.compute_grid()
        logger.info("Compute Elevation")
        self.process_elevation()
        logger.info("Compute Aspect and Slope")
        self.process_aspect_slope()
        logger.info("Compute UCA")
        self.process_uca()
        logger.info("Compute UCA Corrections")
        self.process_uca_edges()
        if self.checkpoint:
            for i in self._owned():              # the state of the fix-up (the reference's uca_edges / edge_* stores, :279-281)
                if not self._stored(i, 'twi'):
                    first = self._load(i, 'uca')
                    self._store(i, None, {'uca_edges': self.tiles[i].uca - first, 'edge_todo': self.tiles[i].edge_todo,
                                          'edge_done': self.tiles[i].edge_done})
        for i in self._owned():
            dp = self.tiles[i]
            if self._stored(i, 'twi'):
                dp.twi = self._load(i, 'twi')
                continue
            dp.find_flats()               # the reference's calc_twi worker rebuilds flats from slope == -1 (:310)
            # ... on a fresh DEMProcessor (:296-307): twi_min_area is the caller's value (default inf), not the smallest
            # cell area calc_uca found on this tile -- it matters for apply_twi_limits / apply_twi_limits_on_uca
            dp.twi_min_area = self.dem_proc_kwargs.get('twi_min_area', np.inf)
            getattr(dp, 'run_twi', dp.calc_twi)()
            if self.checkpoint:
                self._store(i, 'twi', {'twi': dp.twi})
        return [1] * self.n_inputs

    # ------------------------------------------------------------------ results
    def tile_result(self, i, key):
        """Per-tile array under the reference's store names: elev, aspect, slope, uca (first pass),
        uca_edges (edge corrections), edge_todo, edge_done, twi."""
        dp = self.tiles[i]
        if key == 'elev':
            return np.asarray(dp.elev, float)
        if key == 'aspect':
            return dp.direction
        if key == 'slope':
            return dp.mag
        if key == 'uca':
            return dp.uca if self.uca0[i] is None else self.uca0[i]
        if key == 'uca_edges':
            return np.zeros(dp.uca.shape) if self.uca0[i] is None else dp.uca - self.uca0[i]
        if key == 'uca_total':
            return dp.uca
        if key in ('edge_todo', 'edge_done'):
            return getattr(dp, key)
        if key == 'twi':
            return dp.twi
        raise KeyError(key)

    def save_non_overlap_data(self, keys=('elev', 'uca', 'aspect', 'slope', 'twi')):
        """Stitch the uniquely-owned part of every tile into one array per key (reference :742-766;
        'uca' includes the edge corrections as there)."""
        out = {}
        for key in keys:
            full = np.zeros(self.grid_size_tot_unique, self.dtype)
            for i in self._owned():
                su, sl = self.grid_slice_unique[i], self.grid_slice[i]
                loc = (slice(su[0].start - sl[0].start, su[0].stop - sl[0].start),
                       slice(su[1].start - sl[1].start, su[1].stop - sl[1].start))
                src = self.tile_result(i, 'uca_total' if key == 'uca' else key)
                full[self.grid_slice_noverlap[i]] = src[loc]
            out[key] = full
        self.out_file_noverlap = out
        return out

    def save_non_overlap_data_geotiff(self, dtype, crs=None, new_path=None, keys=('elev', 'uca', 'aspect', 'slope', 'twi'), chunks=None,
                                      overview_type=None, overview_factors=None, rescale=None):
        """One GeoTIFF per key, '<new_path>/<key>.tiff', of the stitched non-overlap arrays (reference :786-860; 'uca'
        includes the edge corrections).  Same geotransform, rescaling and overview options as `save_geotiff`; the block
        size `chunks` of the reference's tiled BigTIFF has no meaning for the single-strip files written here."""
        if new_path is None:
            new_path = str(getattr(self, 'out_path', None) or self.in_path).replace('.zarr', '')
        os.makedirs(new_path, exist_ok=True)
        have = getattr(self, 'out_file_noverlap', None) or {}
        if any(k not in have for k in keys):
            self.save_non_overlap_data(keys=tuple(keys))
        for key in keys:
            self.save_geotiff(os.path.join(new_path, key + '.tiff'), key, dtype, crs=crs, rescale=rescale,
                              overview_type=overview_type, overview_factors=overview_factors)

    def process_overviews(self, out_path=None, keys=('elev', 'uca', 'aspect', 'slope', 'twi'), overviews=(3, 3 ** 2, 3 ** 3, 3 ** 4, 3 ** 5, 3 ** 6, 3 ** 7)):
        """The overview pyramid of the stitched results (reference :933-991 with calc_overview :317-352): level `ov` is the
        block mean of the previous level by the factor between them, named '<key>_<ov>'; a key's pyramid ends before the
        first level that would have a side of <= factor cells.  The reference walks its zarr store chunk by chunk; here a
        level is one array = one chunk (`raster.block_mean_overview`, bit-identical to calc_overview on that chunk).
        Returns {name: array}; with `out_path` every level is also written as '<name>.npy' there."""
        from . import raster
        if not getattr(self, 'out_file_noverlap', None):
            self.save_non_overlap_data(keys=[k for k in keys])
        out = {}
        for key in keys:
            last, last_ov = np.asarray(self.out_file_noverlap[key], np.float64), 1
            for ov in overviews:
                factor = ov // last_ov
                new_shape = [-(-n // factor) for n in last.shape]
                if any(n <= factor for n in new_shape):
                    break
                last = raster.block_mean_overview(last, factor).astype(self.dtype)
                last_ov = ov
                out['%s_%d' % (key.split('_')[0], ov)] = last
        if out_path is not None:
            os.makedirs(out_path, exist_ok=True)
            for name, arr in out.items():
                np.save(os.path.join(out_path, name + '.npy'), arr)
        self.overviews = out
        return out

    def save_geotiff(self, filename, key, dtype, crs=None, max_files=2, rescale=None, overview_type=None, overview_factors=None,
                     blocksize=512, bigtiff=True, nodata=None, compress='lzw'):
        """One stitched result as a GeoTIFF (reference :862-931): same geotransform rules (one pixel size for the whole
        mosaic, else NotImplementedError), same optional rescaling, same file layout as the reference's rasterio call --
        512 x 512 tiles in a BigTIFF (`blocksize=None` / `bigtiff=False`: one strip, classic TIFF), LZW compressed like
        the reference's (:905; `compress`: 'lzw', 'deflate' or None).  `overview_type`: one of raster.OVERVIEW_KINDS ('average', 'nearest', 'mode',
        'max', 'min', 'med', 'q1', 'q3', 'sum', 'rms'; GDAL's interpolating kernels are not reproduced) adds reduced-resolution
        images (default factors 3, 9, ... like :928-929) behind the full one; the kind is recorded like :931.
        `crs`: 'projected' or anything else = geographic WGS-84 (the default follows the first input tile)."""
        from . import raster
        if overview_type is not None and overview_type not in raster.OVERVIEW_KINDS:
            raise NotImplementedError("overview_type %r (available: %s)" % (overview_type, ', '.join(raster.OVERVIEW_KINDS)))
        data = self.out_file_noverlap[key]
        dlats = np.unique(np.round(self.index[self.grid_id2i.max(axis=1), 5], decimals=6))
        dlons = np.unique(np.round(self.index[self.grid_id2i.max(axis=0), 4], decimals=6))
        if dlats.size > 1 or dlons.size > 1:
            raise NotImplementedError
        top, bottom = self.index[:, 3].max(), self.index[:, 1].min()
        left, right = self.index[:, 0].min(), self.index[:, 2].max()
        dlat = (bottom - top) / self.grid_size_tot_unique[0]
        dlon = (right - left) / self.grid_size_tot_unique[1]
        if crs is None:
            meta = self._tile_meta[0]
            crs = 'projected' if meta.get('is_projected', 'dX' not in meta) else 'geographic'
        if rescale:
            data = (data - rescale[0]) / (rescale[1] - rescale[0]) * rescale[2]
        levels = []
        if overview_type is not None:
            if overview_factors is None:
                overview_factors = [3 ** i for i in range(1, int(np.log(max(self.grid_size_tot_unique)) / np.log(3)))]   # :928-929
            chain = overview_type in ('average', 'max', 'min', 'sum')          # kinds whose level k is the same statistic of level k-1
            last, last_f = np.asarray(data, np.float64), 1
            for fct in overview_factors:
                if chain and fct % last_f == 0 and fct > last_f:
                    last = raster.block_overview(last, fct // last_f, overview_type)
                else:
                    last = raster.block_overview(np.asarray(data, np.float64), fct, overview_type)
                last_f = fct
                levels.append(last)
        tags = {}
        if rescale:
            tags['rescale'] = ','.join(str(r) for r in rescale)                # (:925; the reference's own call fails on rescale=None)
        if overview_type is not None:
            tags['rio_overview_resampling'] = overview_type                    # :931
        raster.write_geotiff(filename, np.asarray(data).astype(dtype), (dlon, 0.0, left, 0.0, dlat, top),
                             projected=(crs == 'projected'), compress=compress, overviews=levels, tile=blocksize, bigtiff=bigtiff,
                             nodata=nodata, tags=tags or None)
