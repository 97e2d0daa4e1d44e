"""A DEMProcessor look-alike backed by the CPU oracle.  TEST INFRASTRUCTURE ONLY: lets the
ProcessManager's host logic (grid bookkeeping, overlap patching, strip routing, edge rounds) run in
the CPU-only test tier; the product never imports this."""
import numpy as np

from oracle import oracle as O


class OracleProcessor(object):
    def __init__(self, elev=None, dX=None, dY=None, dX2=None, dY2=None, mag=None, direction=None, device=0, **kw):
        self._elev_in = np.asarray(elev)
        self.elev = np.ascontiguousarray(elev, np.float64)
        self.dX, self.dY, self.dX2, self.dY2 = O.spacing_arrays(self.elev.shape[0], dX, dY, dX2, dY2)
        self.fill_flats = kw.get('fill_flats', True)
        self.drain_pits_path = kw.get('drain_pits_path', True)
        self.drain_pits = kw.get('drain_pits', True)
        self.mag = None if mag is None else np.array(mag, float)
        self.direction = None if direction is None else np.array(direction, float)
        self.flats = self.uca = self.twi = self.edge_todo = self.edge_done = None
        self.twi_min_area = kw.get('twi_min_area', np.inf)
        self.twi_min_slope = kw.get('twi_min_slope', 1e-3)
        self.uca_saturation_limit = kw.get('uca_saturation_limit', 32.0)
        self.apply_twi_limits = kw.get('apply_twi_limits', False)
        self.apply_twi_limits_on_uca = kw.get('apply_twi_limits_on_uca', False)
        self._graph = None
        self._pits = []

    # conditioning is host-side product code (pydem_amd/conditioning.py), pinned by its own golden test
    def calc_fill_flats(self):
        from pydem_amd import conditioning
        self.elev = np.ascontiguousarray(conditioning.fill_flats(self._elev_in), np.float64)

    def calc_pit_drain_paths(self):
        from pydem_amd import conditioning
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            self.elev = np.ascontiguousarray(conditioning.pit_drain_paths(np.array(self.elev), self.dX, self.dY)[0], np.float64)

    def calc_slopes_directions(self):
        self.mag, self.direction = O.slopes_directions(self.elev, self.dX, self.dY)
        self.flats = O.flats_edges(self.elev, self.mag, self.direction).astype(bool)
        self._graph = None
        return self.mag, self.direction

    def find_flats(self):
        self.flats = self.mag == -1

    def _build_graph(self):
        flats = np.ascontiguousarray(self.flats, np.uint8)
        mag = np.ascontiguousarray(self.mag)
        self.section, self.proportion = O.section_proportion(np.ascontiguousarray(self.direction), flats, self.dX, self.dY)
        if self.drain_pits:
            pi, pj, pp, _ = O.pit_edges(self.elev, flats, mag, self.dX, self.dY)
            self._pits = np.unique(pi)
        else:
            pi = pj = pp = None
        self.mag = mag
        self.flats = flats.astype(bool)
        self._A = O.adjacency(self.section, self.proportion, self.elev, pi, pj, pp)
        self._graph = True

    def calc_uca(self, uca_init=None, edge_init_data=None):
        n, m = self.elev.shape
        if uca_init is None:
            self._build_graph()
            uca = np.empty((n, m)); todo = np.empty((n, m), np.uint8); done = np.empty((n, m), np.uint8)
            stats = np.zeros(4)
            indptr, indices, data = self._A
            O.lib().oracle_uca_chunk(self.elev, self.section, np.ascontiguousarray(self.flats, np.uint8), n, m, self.dX2, self.dY2,
                                     indptr, indices if indices.size else np.zeros(1, np.int32),
                                     data if data.size else np.zeros(1), 50, 0, 32.0, uca, todo, done, stats)
            self.twi_min_area = min(self.twi_min_area, stats[3])
            self.uca, self.edge_todo, self.edge_done = uca, todo.astype(bool), done.astype(bool)
            return self.uca
        if self._graph is None:
            self._build_graph()
        data, done, todo = edge_init_data
        self.uca, self.edge_todo, self.edge_done = O.uca_update(self.elev, self.flats, self._A, data, done, todo, uca_init)
        return self.uca

    def restore_pit_slopes(self):
        if len(self._pits):
            self.mag.ravel()[self._pits] = -1.0

    def calc_twi(self):
        t = O.twi(self.uca, self.mag, self.twi_min_slope, self.twi_min_area, self.uca_saturation_limit,
                  self.apply_twi_limits, self.apply_twi_limits_on_uca)
        self.twi = t * 10
        return t

    def get_line(self, name, axis, index):
        a = getattr(self, name)
        return np.array(a[index, :] if axis == 0 else a[:, index])

    def set_line(self, name, axis, index, values):
        a = getattr(self, name)
        if axis == 0:
            a[index, :] = values
        else:
            a[:, index] = values
        if name in ('mag', 'direction'):
            self._graph = None
