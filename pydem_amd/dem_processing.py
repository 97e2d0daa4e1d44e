"""Drop-in `DEMProcessor` for the per-tile hot path, backed by the HIP library.

Mirrors the public surface of the reference class (creare-com/pydem v1.2.1,
pydem/dem_processing.py:98-258): same option names and defaults (:105-154), same constructor
normalisation of scalar `dX`/`dY` (:229-242), same methods and result attributes
(`mag, direction, flats, uca, twi, edge_todo, edge_done, section, proportion, twi_min_area`)
and the same quirks (`calc_twi()` returns the un-scaled value while `self.twi` is x10, :1674-1677).
The reference declares its options with traitlets; that package is not a dependency here, the
options are plain attributes.

All arithmetic runs on the GPU through the C-ABI (include/pydem_hip.h); arrays are uploaded once,
stay resident, and are downloaded lazily the first time the corresponding attribute is read.
"""
import logging
import warnings

import numpy as np

from . import _ffi

logger = logging.getLogger(__name__)

FLAT_ID = np.nan
FLAT_ID_INT = -1

_FIELD_OF = {'elev': _ffi.ELEV, 'mag': _ffi.MAG, 'direction': _ffi.DIRECTION, 'flats': _ffi.FLATS,
             'section': _ffi.SECTION, 'proportion': _ffi.PROPORTION, 'uca': _ffi.UCA,
             'edge_todo': _ffi.EDGE_TODO, 'edge_done': _ffi.EDGE_DONE}
_BOOL_FIELDS = ('flats', 'edge_todo', 'edge_done')


class _Resident(object):
    """Attribute whose value lives on the device until somebody reads it."""

    def __init__(self, name):
        self.name = name

    def __get__(self, obj, objtype=None):
        if obj is None:
            return self
        if self.name not in obj._host and self.name in obj._on_device:
            arr = obj._tile.download(_FIELD_OF[self.name])
            if self.name in _BOOL_FIELDS:
                arr = arr.astype(bool)
            obj._host[self.name] = arr
        return obj._host.get(self.name)

    def __set__(self, obj, value):
        obj._on_device.discard(self.name)
        if value is None:
            obj._host.pop(self.name, None)
        else:
            if np.ma.isMaskedArray(value):
                # masked cells are no-data: NaN, like the reference's np.ma.filled(elev.astype(float64), nan) (:213-214)
                value = np.ma.filled(value.astype(np.float64), np.nan)
            obj._host[self.name] = np.asarray(value)


class DEMProcessor(object):
    """Slope magnitude, D-infinity direction, upstream contributing area and TWI of one DEM tile."""

    # ---- options: names and defaults of the reference (dem_processing.py:105-154) ----
    fill_flats = True
    fill_flats_below_sea = False
    fill_flats_source_tol = 1
    fill_flats_peaks = True
    fill_flats_pits = True
    fill_flats_max_iter = 10

    drain_pits = True
    drain_pits_path = True
    drain_pits_min_border = False
    drain_pits_spill = False
    drain_flats = False
    drain_pits_max_iter = 300
    drain_pits_max_dist = 32
    drain_pits_max_dist_XY = None

    apply_uca_limit_edges = False
    apply_twi_limits = False
    apply_twi_limits_on_uca = False

    plotflag = False
    _elev_dtype_after = None
    uca_saturation_limit = 32.0
    twi_min_slope = 1e-3
    twi_min_area = np.inf
    circular_ref_maxcount = 50
    maximum_pit_area = 32.0

    # facet geometry, identical to the reference tables (:173-193)
    facets = [
        [(0, 0), (0, 1), (-1, 1)],
        [(0, 0), (-1, 0), (-1, 1)],
        [(0, 0), (-1, 0), (-1, -1)],
        [(0, 0), (0, -1), (-1, -1)],
        [(0, 0), (0, -1), (1, -1)],
        [(0, 0), (1, 0), (1, -1)],
        [(0, 0), (1, 0), (1, 1)],
        [(0, 0), (0, 1), (1, 1)],
    ]
    ang_adj = np.array([[0, 1], [1, -1], [1, 1], [2, -1], [2, 1], [3, -1], [3, 1], [4, -1]])

    _OPTION_NAMES = ('fill_flats', 'fill_flats_below_sea', 'fill_flats_source_tol', 'fill_flats_peaks',
                     'fill_flats_pits', 'fill_flats_max_iter', 'drain_pits', 'drain_pits_path',
                     'drain_pits_min_border', 'drain_pits_spill', 'drain_flats', 'drain_pits_max_iter',
                     'drain_pits_max_dist', 'drain_pits_max_dist_XY', 'apply_uca_limit_edges',
                     'apply_twi_limits', 'apply_twi_limits_on_uca', 'plotflag', 'uca_saturation_limit',
                     'twi_min_slope', 'twi_min_area', 'circular_ref_maxcount', 'maximum_pit_area',
                     'bounds', 'transform', 'done')

    elev = _Resident('elev')
    mag = _Resident('mag')
    direction = _Resident('direction')
    flats = _Resident('flats')
    section = _Resident('section')
    proportion = _Resident('proportion')
    uca = _Resident('uca')
    edge_todo = _Resident('edge_todo')
    edge_done = _Resident('edge_done')

    @property
    def twi(self):
        """10 x ln(uca / (mag + min_slope)) (the reference stores the scaled value, :1674); downloaded lazily."""
        if self._twi10 is None and self._twi_on_device:
            self._twi10 = self._tile.download(_ffi.TWI) * 10
        return self._twi10

    @twi.setter
    def twi(self, value):
        self._twi10 = None if value is None else np.asarray(value)
        self._twi_on_device = False

    def __init__(self, elev_fn=None, device=0, **kwargs):
        if elev_fn:
            # reference :229-232: the raster's array, spacing, bounds and transform, overridden by explicit keywords
            from .raster import dem_processor_from_raster_kwargs
            kwds = dem_processor_from_raster_kwargs(elev_fn)
            kwds.update(kwargs)
            kwargs = kwds
        self._shape = None
        if kwargs.get('elev') is None:
            if kwargs.get('shape') is None:
                raise ValueError("DEMProcessor needs an elevation array (elev=...)")
            self._shape = tuple(int(v) for v in kwargs.pop('shape'))     # elevation will be produced on the device
            kwargs.pop('elev', None)
            n_rows = self._shape[0]
        else:
            n_rows = np.shape(kwargs['elev'])[0]
        # scalar / missing spacing -> per-row arrays (reference :233-240, defaults :244-258)
        if not isinstance(kwargs.get('dX'), np.ndarray):
            if 'dX2' not in kwargs:
                kwargs['dX2'] = np.ones(n_rows) * kwargs.get('dX', 1)
            kwargs['dX'] = np.ones(n_rows - 1) * kwargs.get('dX', 1)
        if not isinstance(kwargs.get('dY'), np.ndarray):
            if 'dY2' not in kwargs:
                kwargs['dY2'] = np.ones(n_rows) * kwargs.get('dY', 1)
            kwargs['dY'] = np.ones(n_rows - 1) * kwargs.get('dY', 1)
        self._host = {}
        self._on_device = set()
        self._uploaded = set()
        self._tile = None
        self._device = device
        self._twi10 = None
        self._twi_on_device = False
        self.A = None
        self.bounds = []
        self.transform = []
        self.done = None
        self.dX = np.array(kwargs.pop('dX'), dtype='float64')
        self.dY = np.array(kwargs.pop('dY'), dtype='float64')
        self.dX2 = np.array(kwargs.pop('dX2', np.ones(n_rows)), dtype='float64')
        self.dY2 = np.array(kwargs.pop('dY2', np.ones(n_rows)), dtype='float64')
        for nm in ('dX', 'dY'):
            d = getattr(self, nm)
            if not (np.isfinite(d).all() and (d > 0).all()):
                # (the reference runs on with negative / zero cell sizes and returns mirrored / infinite slopes; the device
                # kernels compare slopes by cross-multiplication with the spacings and refuse such input instead)
                raise ValueError("%s must be finite and > 0 (pass cell sizes, not signed geotransform steps)" % nm)
        area = self.dX2 * self.dY2
        if not (np.isfinite(area).all() and (area > 0).all()):
            raise ValueError("dX2 * dY2 (the cell areas) must be finite and > 0")   # the sweep keeps a flag in the sign of a share
        for k, v in kwargs.items():
            if k in _FIELD_OF:
                setattr(self, k, v)
            elif k == 'twi':
                self.twi = v
            elif k in self._OPTION_NAMES:
                setattr(self, k, v)
            elif k in ('bounds', 'transform'):
                setattr(self, k, list(v))              # tl.List() traits of the reference (:201-202)
            # unknown keywords are dropped, as traitlets' HasTraits.__init__ does

    # ------------------------------------------------------------------ device plumbing
    def _ensure_tile(self):
        if self._tile is None:
            n, m = self._shape if self._shape is not None else self.elev.shape
            self._tile = _ffi.Tile(n, m, self._device)
        # the reference lets callers overwrite dX/dY after construction (process_manager.py:59-63): re-send them
        # whenever one of the four attributes was rebound (an edge round calls this ~100 times per tile)
        cur = (self.dX, self.dY, self.dX2, self.dY2)
        last = getattr(self, '_spacing_sent', None)
        if last is None or any(a is not b for a, b in zip(cur, last)):
            self._tile.set_spacing(*cur)
            self._spacing_sent = cur
        return self._tile

    def _push(self, *names):
        """Upload host-side fields that the device does not hold yet."""
        tile = self._tile
        for nm in names:
            if nm in self._on_device:
                continue
            arr = self._host.get(nm)
            if arr is None:
                raise RuntimeError("field %r is required but has not been set or computed" % nm)
            tile.upload(_FIELD_OF[nm], arr)
            self._on_device.add(nm)

    def _produced(self, *names):
        for nm in names:
            self._host.pop(nm, None)
            self._on_device.add(nm)

    def _options(self):
        o = _ffi.Options()
        o.drain_pits = int(bool(self.drain_pits))
        o.drain_pits_min_border = int(bool(self.drain_pits_min_border))
        o.drain_pits_max_iter = int(self.drain_pits_max_iter)
        o.drain_pits_max_dist = int(self.drain_pits_max_dist or 0)
        o.drain_pits_max_dist_XY = float(self.drain_pits_max_dist_XY) if self.drain_pits_max_dist_XY else float('nan')
        o.apply_uca_limit_edges = int(bool(self.apply_uca_limit_edges))
        o.apply_twi_limits = int(bool(self.apply_twi_limits))
        o.apply_twi_limits_on_uca = int(bool(self.apply_twi_limits_on_uca))
        o.circular_ref_maxcount = int(self.circular_ref_maxcount)
        o.uca_saturation_limit = float(self.uca_saturation_limit)
        o.twi_min_slope = float(self.twi_min_slope)
        o.twi_min_area = float(self.twi_min_area)
        return o

    def _has(self, name):
        return name in self._on_device or self._host.get(name) is not None

    # one row (axis 0) / column (axis 1) of a field without moving the whole array off the device
    def get_line(self, name, axis, index):
        if name in self._on_device and name not in self._host:
            arr = self._tile.get_line(_FIELD_OF[name], axis, index)
            return arr.astype(bool) if name in _BOOL_FIELDS else arr
        a = self._host[name]
        return np.array(a[index, :] if axis == 0 else a[:, index])

    def get_lines(self, requests):
        """[(name, axis, index)] -> list of 1-D arrays; device-resident fields are fetched with one synchronisation."""
        dev = [k for k, (name, _, _) in enumerate(requests) if name in self._on_device and name not in self._host]
        out = [None] * len(requests)
        if len(dev) > 1:
            got = self._tile.get_lines([(_FIELD_OF[requests[k][0]], requests[k][1], requests[k][2]) for k in dev])
            for k, arr in zip(dev, got):
                out[k] = arr.astype(bool) if requests[k][0] in _BOOL_FIELDS else arr
        for k, (name, axis, index) in enumerate(requests):
            if out[k] is None:
                out[k] = self.get_line(name, axis, index)
        return out

    def set_line(self, name, axis, index, values):
        if name in self._on_device:
            self._tile.set_line(_FIELD_OF[name], axis, index, values)
            self._host.pop(name, None)
        else:
            a = self._host[name]
            if axis == 0:
                a[index, :] = values
            else:
                a[:, index] = values

    @property
    def timings(self):
        return self._tile.timings() if self._tile is not None else {}

    @classmethod
    def from_synthetic(cls, shape, synth, device=0, **kwargs):
        """Tile whose elevation is generated on the device by the deterministic fractal generator
        (pydem_amd/synth.py:fractal parameters) -- bench / test input, never leaves HBM."""
        dp = cls(shape=shape, device=device, **kwargs)
        dp._ensure_tile()
        dp._tile.synth_fractal(**synth)
        dp._on_device.add('elev')
        return dp

    @property
    def shape(self):
        # never a reason to bring the elevation back from the device (a 8192^2 float64 surface is 512 MB over PCIe)
        if self._shape is not None:
            return self._shape
        h = self._host.get('elev')
        if h is not None:
            return tuple(np.shape(h))
        if self._tile is not None:
            return tuple(self._tile.shape)
        return self.elev.shape

    # ------------------------------------------------------------------ reference API
    def find_flats(self):
        """flats = (mag == -1)  (reference :305-306)"""
        self._ensure_tile()
        self._push('mag')
        self._tile.find_flats()
        self._produced('flats')

    # ---- elevation conditioning: artefacts, flats and pit drain paths on the device (csrc/cond_device.hip, csrc/cond_paths.hip),
    # tiles with no-data cells included; masked arrays / exotic dtypes go through the host implementation (pydem_amd/conditioning.py)
    def _condition_on_device(self, artefacts_only):
        """True when the resident elevation was conditioned by the library (False: the caller falls back to the host)."""
        elev = self._host.get('elev')
        if elev is not None and (np.ma.isMaskedArray(elev) or np.asarray(elev).dtype.kind not in 'iuf' or np.asarray(elev).ndim != 2):
            return False
        if min(self.shape) < 3:
            return False
        self._ensure_tile()
        self._push('elev')
        ok = self._tile.fill_flats(self.maximum_pit_area if (self.maximum_pit_area or artefacts_only) else 0.0, self.fill_flats_below_sea,
                                   self.fill_flats_source_tol, self.fill_flats_peaks, self.fill_flats_pits, artefacts_only)
        if not ok:
            return False
        dtype = None if elev is None else np.asarray(elev).dtype
        self._produced('elev')
        for nm in ('mag', 'direction', 'flats', 'uca', 'section', 'proportion', 'edge_todo', 'edge_done'):
            self._on_device.discard(nm)             # the conditioning used those planes as scratch
        self._elev_dtype_after = dtype if artefacts_only else None
        return True

    def calc_fill_pit_artifacts(self):
        """Fill quantisation pits (reference :396-426)."""
        if self._condition_on_device(artefacts_only=True):
            if self._elev_dtype_after is not None and self._elev_dtype_after != np.float64:
                self.elev = self.elev.astype(self._elev_dtype_after)          # the step keeps the array's dtype
            return
        from . import conditioning
        self.elev = conditioning.fill_pit_artifacts(self.elev, self.maximum_pit_area, self.fill_flats_below_sea)

    def calc_fill_flats(self):
        """Fill / interpolate flats before the slope stencil (reference :551-579)."""
        if self._condition_on_device(artefacts_only=False):
            return
        from . import conditioning
        self.elev = conditioning.fill_flats(self.elev, self.maximum_pit_area, self.fill_flats_below_sea,
                                            self.fill_flats_source_tol, self.fill_flats_peaks, self.fill_flats_pits)

    def calc_pit_drain_paths(self):
        """Carve monotone paths from pits to their outlets (reference :428-548).  Works on a copy of the
        array (the reference edits the caller's array in place).  Returns the drained surface like the reference;
        `run_pit_drain_paths` is the same step without bringing it back to the host."""
        self.run_pit_drain_paths()
        return self.elev

    def run_pit_drain_paths(self):
        res = self._pit_paths_on_device()
        if res is None and self._tile is not None and 'elev' in self._on_device:
            warnings.warn("calc_pit_drain_paths: the device schedule handed this tile to the sequential host loop "
                          "(a conflict of the speculative rounds): same result, much slower")
        if res is not None:
            n_failed, used, self._pit_path_rounds = res
            if n_failed:
                warnings.warn("Warning %d pits had no place to drain to in this chunk" % n_failed)
            logger.info("... done draining pits with maxiter = %d", used)
            return
        from . import conditioning
        elev = np.array(self.elev)
        elev, n_failed, used = conditioning.pit_drain_paths(elev, self.dX, self.dY, self.drain_pits_max_iter,
                                                            self.drain_pits_max_dist, self.drain_pits_max_dist_XY,
                                                            self.fill_flats_below_sea)
        logger.info("... done draining pits with maxiter = %d", used)
        self.elev = elev

    def _pit_paths_on_device(self):
        """(n_failed, iterations, rounds) when the library carved the paths on the resident surface, else None (tiles with
        no-data cells, dtypes the device cannot hold, or the parallel schedule gave up: the caller runs the host loop).
        Integer / float32 surfaces keep numpy's in-dtype arithmetic (:537-539): the library truncates / rounds the path
        values like the array's dtype would, the pits are sorted on keys of that dtype, and the surface comes back in it."""
        elev = self._host.get('elev')
        dtype = None
        if elev is not None:
            if np.ma.isMaskedArray(elev) or np.asarray(elev).ndim != 2:
                return None
            dtype = np.asarray(elev).dtype
            if dtype not in (np.dtype('float64'), np.dtype('float32'), np.dtype('int16'), np.dtype('int32'), np.dtype('uint8'), np.dtype('int8')):
                return None
        if min(self.shape) < 3:
            return None
        self._ensure_tile()
        self._push('elev')
        res = self._tile.pit_drain_paths(self.fill_flats_below_sea, self.drain_pits_max_iter, self.drain_pits_max_dist,
                                         self.drain_pits_max_dist_XY, sort_dtype=dtype)
        if res is None:
            return None
        self._produced('elev')
        for nm in ('mag', 'direction', 'flats', 'uca', 'section', 'proportion', 'edge_todo', 'edge_done'):
            self._on_device.discard(nm)
        if dtype is not None and dtype != np.float64:
            self.elev = self.elev.astype(dtype)                # the reference edits the array in place: it keeps its dtype
        return res

    def calc_slopes_directions(self, plotflag=False):
        """Slope magnitude and D-infinity direction (reference :587-619)."""
        self.run_slopes_directions()
        return self.mag, self.direction

    def run_slopes_directions(self):
        """calc_slopes_directions without bringing the results back to the host."""
        if self.fill_flats:
            self.calc_fill_flats()
        if self.drain_pits_path:
            self.run_pit_drain_paths()
        self._ensure_tile()
        self._push('elev')
        logger.info("Starting slope/direction calculation")
        self._tile.slopes_directions()
        self._produced('mag', 'direction', 'flats')

    def calc_uca(self, plotflag=False, edge_init_data=None, uca_init=None):
        """Upstream contributing area (reference :682-776)."""
        self.run_uca(edge_init_data=edge_init_data, uca_init=uca_init)
        return self.uca

    def run_uca(self, edge_init_data=None, uca_init=None, uca_resident=False, incremental=False):
        """calc_uca without bringing the result back to the host.  `uca_resident=True` (edge rounds only)
        says uca_init is the tile's own device-resident UCA, so nothing is uploaded; `incremental=True` (with
        uca_resident) runs the round on the persistent fix-up state (pydem_uca_edge_round_inc): finished cells and
        masks are those of the plain round, cells below an unresolved inlet are settled by `flush_edge_rounds()`."""
        if not self._has('direction'):
            self.run_slopes_directions()
        if uca_init is not None or uca_resident:
            return self._calc_uca_edge_round(uca_init, edge_init_data, uca_resident,
                                             incremental and uca_resident and not self.apply_uca_limit_edges)
        if not self.drain_pits and (self.drain_flats or self.drain_pits_spill):
            # (_mk_connectivity_flats / _mk_connectivity_pits_spill, dem_processing.py:1108-1123: alternatives the
            # reference itself labels "not a great option"; only reachable with drain_pits=False)
            raise NotImplementedError("drain_flats / drain_pits_spill (without drain_pits) are not implemented on the "
                                      "device path; use drain_pits=True (the reference default) or leave both off")
        self._ensure_tile()
        self._push('elev', 'mag', 'direction', 'flats')
        opt = self._options()
        logger.info("Starting uca calculation")
        self._tile.uca(opt)
        tm = self._tile.timings()
        if tm['n_unresolved']:
            # circular drainage that the re-seed loop (dem_processing.py:951-964, replayed on the device) did not resolve
            # within circular_ref_maxcount rounds: like the reference, those cells keep the area they have received
            warnings.warn("%d cells lie on or below a circular drainage pattern that the re-seed loop did not resolve"
                          % tm['n_unresolved'])
        if tm['n_pits_undrained']:
            warnings.warn("Warning %d pits had no place to drain to in this chunk" % tm['n_pits_undrained'])
        self.twi_min_area = min(self.twi_min_area, opt.twi_min_area)
        # pits that found a drain are patched into mag/flats by the graph stage (reference :1369-1371)
        self._produced('section', 'proportion', 'uca', 'edge_todo', 'edge_done', 'mag', 'flats')

    def build_graph(self):
        """The flow graph for a tile whose slope / aspect were set instead of computed (a resumed directory job): built now,
        before stored edge masks are uploaded (the graph stage resets them)."""
        self._ensure_tile()
        if not self._has('flats'):
            self.find_flats()
        self._push('elev', 'mag', 'direction', 'flats')
        self._tile.build_graph(self._options())
        self._produced('mag', 'flats', 'section', 'proportion')

    def restore_pit_slopes(self):
        """mag = -1 again at the pits drained by calc_uca (what the reference's slope *store* holds in
        the directory flow, process_manager.py:192-194)."""
        if self._tile is not None and 'mag' in self._on_device:
            self._tile.restore_pit_slopes()
            self._host.pop('mag', None)

    def run_edge_round_dev(self):
        """One incremental edge round whose strips the edge board (pydem_board_eval) has put into the tile's buffers."""
        self._ensure_tile()
        self._tile.uca_edge_round_dev(self._options())
        self._produced('uca', 'edge_todo', 'edge_done')

    def flush_edge_rounds(self):
        """End of a series of incremental edge rounds (no-op otherwise)."""
        if self._tile is not None:
            self._tile.uca_edge_flush()
            self._host.pop('uca', None)

    def _calc_uca_edge_round(self, uca_init, edge_init_data, uca_resident=False, incremental=False):
        """calc_uca(uca_init=..., edge_init_data=[data, done, todo]) of the reference (:724-771):
        only the contributions entering through finished neighbour edges are propagated."""
        keys = ('left', 'right', 'top', 'bottom')
        n, m = self.shape
        lens = dict(left=n, right=n, top=m, bottom=m)
        if edge_init_data is None:
            data = {k: np.zeros(lens[k]) for k in keys}
            done = {k: np.zeros(lens[k], bool) for k in keys}
            todo = {k: np.zeros(lens[k], bool) for k in keys}
        else:
            data, done, todo = edge_init_data
        self._ensure_tile()
        if not self._has('flats'):
            self.find_flats()
        if not (uca_resident and self._has('uca')):                           # (a resumed tile holds its own uca on the host)
            self.uca = np.asarray(uca_init).astype('float64')                 # :744
        self._push('elev', 'mag', 'direction', 'flats', 'uca')
        opt = self._options()
        logger.info("Starting edge resolution round")
        self._tile.uca_edge_update(opt, [data[k] for k in keys], [done[k] for k in keys], [todo[k] for k in keys],
                                   incremental=incremental and 'edge_done' in self._on_device)
        self._produced('uca', 'edge_todo', 'edge_done', 'mag', 'flats', 'section', 'proportion')

    def calc_twi(self):
        """Topographic wetness index; returns the un-scaled array, stores 10x in self.twi (:1647-1677)."""
        self.run_twi()
        return self._tile.download(_ffi.TWI)

    def run_twi(self):
        """calc_twi without bringing the result back to the host."""
        if not self._has('uca'):
            self.run_uca()
        self._ensure_tile()
        self._push('uca', 'mag')
        self._tile.twi(self._options())
        self._twi10 = None
        self._twi_on_device = True
