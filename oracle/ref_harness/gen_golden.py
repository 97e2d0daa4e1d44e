"""Generate golden vectors under tests/golden/ by running the UNMODIFIED reference
(/root/reference, creare-com/pydem v1.2.1) on small deterministic inputs.

Run in the build container only:   oracle/ref_harness/run.sh oracle/ref_harness/gen_golden.py
Fixtures hold data only (inputs + the reference's outputs); no reference source.

Every case records, for one DEMProcessor(elev=..., **kw).calc_twi() run of the reference
(call stack: SURVEY.md section 3.1), the intermediate arrays at the boundaries the
oracle/HIP path reproduce.
"""
import hashlib
import os
import sys
import warnings

from load_reference import load_reference
pydem = load_reference()

import numpy as np  # noqa: E402
from pydem.dem_processing import DEMProcessor  # noqa: E402
from pydem import dem_processing as refmod  # noqa: E402

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, REPO)
from pydem_amd import synth  # noqa: E402

OUT = os.path.join(REPO, 'tests', 'golden')


def run_case(elev, dX=None, dY=None, **kw):
    """Run the reference end to end, capturing intermediates by wrapping its methods."""
    rec = {}
    kwargs = dict(kw)
    if dX is not None:
        kwargs['dX'] = dX
    if dY is not None:
        kwargs['dY'] = dY
    dp = DEMProcessor(elev=elev.copy(), **kwargs)
    rec['in_elev'] = elev.copy()
    rec['in_dX'] = np.array(dp.dX, 'float64')
    rec['in_dY'] = np.array(dp.dY, 'float64')
    rec['in_dX2'] = np.array(dp.dX2, 'float64')
    rec['in_dY2'] = np.array(dp.dY2, 'float64')

    def wrap(name, post):
        orig = getattr(dp, name)

        def f(*a, **k):
            r = orig(*a, **k)
            post(r, a, k)
            return r
        # instance attribute shadows the class method
        object.__setattr__(dp, name, f)

    wrap('calc_fill_pit_artifacts', lambda r, a, k: rec.__setitem__('elev_artifacts', np.array(dp.elev).copy()))
    wrap('calc_fill_flats', lambda r, a, k: rec.__setitem__('elev_filled', np.array(dp.elev).copy()))
    wrap('calc_pit_drain_paths', lambda r, a, k: rec.__setitem__('elev_drained', np.array(dp.elev).copy()))

    def post_sd(r, a, k):
        rec['mag'] = r[0].copy()
        rec['direction'] = r[1].copy()
        rec['flats'] = dp.flats.copy()
    wrap('calc_slopes_directions', post_sd)

    def post_raw(r, a, k):
        rec['mag_raw'] = r[0].copy()
        rec['direction_raw'] = r[1].copy()
    wrap('_slopes_directions', post_raw)

    def post_sp(r, a, k):
        rec['section'] = r[0].copy()
        rec['proportion'] = r[1].copy()
    wrap('_calc_uca_section_proportion', post_sp)

    def post_pits(r, a, k):
        rec['pit_i'], rec['pit_j'], rec['pit_prop'] = [np.array(x) for x in r[:3]]
    wrap('_mk_connectivity_pits', post_pits)

    def post_A(r, a, k):
        A = r.copy()
        A.sort_indices()
        rec['A_indptr'] = A.indptr.copy()
        rec['A_indices'] = A.indices.copy()
        rec['A_data'] = A.data.copy()
    wrap('_mk_adjacency_matrix', post_A)

    calls = []
    orig_da = refmod.cyutils.drain_area

    class _Cy:
        drain_connections = staticmethod(refmod.cyutils.drain_connections)

        @staticmethod
        def drain_area(*a, **k):
            calls.append(1)
            return orig_da(*a, **k)
    saved = refmod.cyutils
    refmod.cyutils = _Cy
    try:
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            twi = dp.calc_twi()
    finally:
        refmod.cyutils = saved
    rec['n_drain_area_calls'] = np.int64(len(calls))
    rec['elev_final'] = np.array(dp.elev).copy()
    rec['mag_final'] = dp.mag.copy()
    rec['flats_final'] = dp.flats.copy()
    rec['uca'] = dp.uca.copy()
    rec['edge_todo'] = dp.edge_todo.copy()
    rec['edge_done'] = dp.edge_done.copy()
    rec['twi_ret'] = twi.copy()
    rec['twi_attr'] = dp.twi.copy()
    rec['twi_min_area'] = np.float64(dp.twi_min_area)
    return rec


def save(name, rec, kw):
    os.makedirs(OUT, exist_ok=True)
    rec = dict(rec)
    rec['kwargs_repr'] = np.array(repr(sorted(kw.items())))
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **rec)
    print('%-34s %8.1f KiB  drain_area calls=%d' % (name, os.path.getsize(path) / 1024., int(rec['n_drain_area_calls'])))


def read_tif32():
    """Decode the reference test fixture pydem/test/test_NN032_033_elev.tif
    (32x32 float64, uncompressed, single strip) with struct -- no rasterio."""
    import struct
    b = open('/root/reference/pydem/test/test_NN032_033_elev.tif', 'rb').read()
    assert b[:2] == b'II'
    off = struct.unpack('<I', b[4:8])[0]
    n = struct.unpack('<H', b[off:off + 2])[0]
    tags = {}
    for k in range(n):
        tag, typ, cnt, val = struct.unpack('<HHII', b[off + 2 + 12 * k: off + 14 + 12 * k])
        tags[tag] = (typ, cnt, val)
    w, h = tags[256][2], tags[257][2]
    assert tags[258][2] == 64 and tags[259][2] == 1 and tags[339][2] == 3
    strip = tags[273][2]
    return np.frombuffer(b[strip:strip + w * h * 8], '<f8').reshape(h, w).copy()


def run_edge_update_case(name, elev, dX, dY, seed, **kw):
    """A9: one edge-resolution round exactly as process_manager.calc_uca_ec drives it
    (process_manager.py:224-284): a FRESH DEMProcessor built from stored elev/aspect/slope,
    find_flats(), then calc_uca(uca_init=..., edge_init_data=[data, done, todo])."""
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        dp0 = DEMProcessor(elev=elev.copy(), dX=dX, dY=dY, fill_flats=False, drain_pits_path=False, **kw)
        mag0, dir0 = dp0.calc_slopes_directions()
        mag0 = mag0.copy(); dir0 = dir0.copy()        # what the aspect/slope workers store (before the pit patch)
        dp0.calc_uca()
        uca0 = dp0.uca.copy(); todo0 = dp0.edge_todo.copy(); done0 = dp0.edge_done.copy()
        rng = np.random.default_rng(seed)
        n, m = elev.shape
        sides = {'left': (slice(None), 0), 'right': (slice(None), -1), 'top': (0, slice(None)), 'bottom': (-1, slice(None))}
        data, dn, td = {}, {}, {}
        for k, sl in sides.items():
            L = uca0[sl].size
            data[k] = np.nan_to_num(uca0[sl], nan=1.0) + rng.random(L) * 50000.0 * (rng.random(L) < 0.7)
            dn[k] = rng.random(L) < 0.6
            td[k] = todo0[sl].copy()
        dp = DEMProcessor(elev=elev.copy(), dX=dX, dY=dY, mag=mag0.copy(), direction=dir0.copy(), fill_flats=False,
                          drain_pits_path=False, **kw)
        dp.find_flats()
        uca1 = dp.calc_uca(uca_init=uca0.copy(), edge_init_data=[{k: v.copy() for k, v in data.items()},
                                                                  {k: v.copy() for k, v in dn.items()},
                                                                  {k: v.copy() for k, v in td.items()}]).copy()
    rec = dict(in_elev=elev, in_dX=np.array(dp.dX, float), in_dY=np.array(dp.dY, float), in_dX2=np.array(dp.dX2, float),
               in_dY2=np.array(dp.dY2, float), in_mag=mag0, in_direction=dir0, uca_init=uca0, edge_todo_init=todo0,
               edge_done_init=done0, uca=uca1, edge_todo=dp.edge_todo.copy(), edge_done=dp.edge_done.copy(),
               n_drain_area_calls=np.int64(0))
    for k in sides:
        rec['strip_data_' + k] = data[k]; rec['strip_done_' + k] = dn[k]; rec['strip_todo_' + k] = td[k]
    save(name, rec, kw)


def conditioning_cases():
    """g7: inputs that exercise every branch of the conditioning code (artefact pits, summit plateaus,
    closed depressions, flats leaving through the tile edge, tiny windows, integer dtype paths)."""
    rng = np.random.default_rng(7)
    # integer terraces with one-unit pits of several sizes, plateaus on peaks and against the edges
    base = np.rint(synth.fractal(72, 80, seed=21, top_shift=5, n_octaves=4, zrange=12.0)).astype(np.int16) + 5
    for _ in range(25):
        i, j = rng.integers(2, 68), rng.integers(2, 76)
        h, w = rng.integers(1, 4), rng.integers(1, 4)
        lvl = base[i - 1:i + h + 1, j - 1:j + w + 1].min()
        base[i - 1:i + h + 1, j - 1:j + w + 1] = lvl + 1
        base[i:i + h, j:j + w] = lvl
    run_and_save('g7_int16_terraces', base, 30.0, 30.0, dict())
    run_and_save('g7_int16_terraces_noff', base, 30.0, 30.0, dict(fill_flats=False))
    # float surface with exact plateaus (lakes), a summit plateau and a plateau cut by the border
    f = synth.fractal(64, 64, seed=22, top_shift=5, n_octaves=5, zrange=80.0)
    f = np.maximum(f, np.quantile(f, 0.2))
    f[5:12, 20:30] = f.max() + 3.0
    f[:4, 40:52] = f[:4, 40:52].min()
    f[30:33, -5:] = f[30:33, -5:].min()
    run_and_save('g7_float_lakes', f, 10.0, 12.0, dict())
    # many tiny features: 2-level random integers
    tiny = rng.integers(1, 4, size=(40, 44)).astype(np.float64)
    run_and_save('g7_tiny_features', tiny, 1.0, 1.0, dict())
    # conditioning with anisotropic, row-varying spacing
    n = 56
    v = np.rint(synth.fractal(n, 48, seed=23, top_shift=4, n_octaves=4, zrange=30.0))
    run_and_save('g7_varspacing', v, 20.0 + 0.1 * np.arange(n - 1), 35.0 - 0.05 * np.arange(n - 1), dict())
    # no-data cells (NaN) in the conditioning: the masks of calc_fill_pit_artifacts / calc_fill_flats / calc_pit_drain_paths go
    # through scipy's minimum_filter, whose ring algorithm lets a NaN shield its neighbours -- regions of mixed height,
    # rims with lower cells, pits next to voids (round 4: the device path replays the filter value for value)
    rn = np.random.default_rng(77)
    g = synth.fractal(72, 88, seed=24, top_shift=5, n_octaves=5, zrange=60.0)
    g = np.maximum(g, np.quantile(g, 0.25))                      # lakes
    g[8:15, 30:44] = g.max() + 2.0                               # a summit plateau
    g[20:34, 50:70] = np.nan                                     # a void that cuts the lake shore
    g[:, :3] = np.nan                                            # a no-data margin
    g[rn.random(g.shape) < 0.01] = np.nan                        # scattered voids
    run_and_save('g7_nan_lakes', g, 10.0, 12.0, dict())
    t2 = np.rint(synth.fractal(64, 70, seed=25, top_shift=5, n_octaves=4, zrange=10.0)) + 3.0
    for _ in range(18):
        i, j = rn.integers(2, 60), rn.integers(2, 66)
        lvl = t2[i - 1:i + 3, j - 1:j + 3].min()
        t2[i - 1:i + 3, j - 1:j + 3] = lvl + 1
        t2[i:i + 2, j:j + 2] = lvl
    t2[rn.random(t2.shape) < 0.04] = np.nan
    t2[40:52, 10:22] = np.nan
    run_and_save('g7_nan_terraces', t2, 30.0, 30.0, dict())
    run_and_save('g7_nan_terraces_opts', t2, 30.0, 30.0, dict(fill_flats_below_sea=True, fill_flats_source_tol=3, maximum_pit_area=4,
                                                              drain_pits_max_iter=40, drain_pits_max_dist=6))
    s2 = synth.fractal(60, 64, seed=26, top_shift=5, n_octaves=5, zmin=-15.0, zrange=60.0)
    s2 = np.rint(s2 * 2) / 2
    s2[s2 <= 0] = np.nan                                          # the sea as no-data
    run_and_save('g7_nan_sea', s2, 25.0, 25.0, dict())


def run_and_save(name, elev, dX, dY, kw):
    save(name, run_case(elev, dX, dY, **kw), kw)


def main():
    if '--only-conditioning' in sys.argv:
        conditioning_cases()
        write_manifest()
        return
    if '--only-edge' in sys.argv:
        edge_cases()
        write_manifest()
        return
    cases = []
    # G1: the reference's own known-answer inputs (test_end_to_end.py:153-158, 221-226)
    card = np.array([[1] * 5, [2] * 5, [3] * 5, [4] * 5, [5] * 5])
    diag = np.add.outer(np.arange(5), np.arange(5)) + 1
    for nm, e in (('g1_cardinal', card), ('g1_diagonal', diag)):
        for sym, f in (('id', lambda a: a), ('rev', lambda a: a[::-1]), ('t', lambda a: a.T),
                       ('trev', lambda a: a[::-1, ::-1].T)):
            cases.append(('%s_%s' % (nm, sym), np.ascontiguousarray(f(e)), None, None, dict(fill_flats=False)))
    # G2: cone
    cases.append(('g2_cone64', synth.cone(64), None, None, dict(fill_flats=False)))
    cases.append(('g2_cone64_nopit', synth.cone(64), None, None,
                  dict(fill_flats=False, drain_pits=False, drain_pits_path=False)))
    cases.append(('g2_conescaled32_dxy', synth.cone_scaled(32), 1.0, 1.0, dict(fill_flats=False)))
    # G3: the reference's GeoTIFF fixture, decoded, dX=dY=1
    cases.append(('g3_tif32', read_tif32(), 1.0, 1.0, dict(fill_flats=False)))
    # G4: fractal fp64, several option sets; non-square pixels, varying per-row spacing
    fr = synth.fractal(96, 128, seed=0, top_shift=6, n_octaves=6)
    cases.append(('g4_fractal_nopits', fr, 30.0, 30.0,
                  dict(fill_flats=False, drain_pits=False, drain_pits_path=False)))
    cases.append(('g4_fractal_pits', fr, 30.0, 30.0, dict(fill_flats=False, drain_pits_path=False)))
    cases.append(('g4_fractal_defaults_noff', fr, 30.0, 30.0, dict(fill_flats=False)))
    n = 80
    fr2 = synth.fractal(n, 72, seed=5, top_shift=5, n_octaves=5, zrange=300.0)
    dXv = 25.0 + 0.05 * np.arange(n - 1)
    dYv = 31.0 - 0.01 * np.arange(n - 1)
    cases.append(('g4_fractal_varspacing', fr2, dXv, dYv, dict(fill_flats=False, drain_pits_path=False)))
    # G5: int16 SRTM-like, all defaults (fill_flats=True)
    sr = synth.srtm_int16(96, 96, seed=3, top_shift=6, n_octaves=6, zrange=400.0, lake_level=150)
    cases.append(('g5_int16_defaults', sr, 30.0, 30.0, dict()))
    cases.append(('g5_int16_noff_nopath', sr, 30.0, 30.0, dict(fill_flats=False, drain_pits_path=False)))
    quant = np.rint(synth.fractal(64, 64, seed=9, top_shift=5, n_octaves=5, zrange=60.0)).astype(np.float64)
    cases.append(('g5_quant_f64_pits', quant, 10.0, 10.0, dict(fill_flats=False, drain_pits_path=False)))

    # G4 with nodata: NaN lake, NaN block on the tile edge, isolated NaN cells (how the reference treats missing data
    # in the stencil, the flats, the graph, the pit search and the edge flags)
    frn = synth.fractal(72, 96, seed=11, top_shift=5, n_octaves=5)
    rng = np.random.default_rng(11)
    frn[20:34, 30:52] = np.nan
    frn[0:9, 70:] = np.nan
    for _ in range(25):
        frn[rng.integers(0, 72), rng.integers(0, 96)] = np.nan
    cases.append(('g4_fractal_nan_pits', frn, 30.0, 30.0, dict(fill_flats=False, drain_pits_path=False)))
    cases.append(('g4_fractal_nan_nopits', frn, 30.0, 30.0, dict(fill_flats=False, drain_pits=False, drain_pits_path=False)))
    # sea: large areas at exactly 0 (SRTM oceans), a trench below 0 -- `elev > 0` gates the pit search (:1284) and the
    # flat filling (:565), so these cells stay flats
    sea = synth.fractal(80, 96, seed=13, top_shift=5, n_octaves=5, zrange=400.0) - 150.0
    sea[sea < 0] = 0.0
    sea[50:56, 10:40] = -12.5
    cases.append(('g4_fractal_sea_pits', sea, 30.0, 30.0, dict(fill_flats=False, drain_pits_path=False)))
    sea16 = np.rint(sea).astype(np.int16)
    cases.append(('g5_int16_sea_defaults', sea16, 30.0, 30.0, dict()))
    # non-default options (every trait of DEMProcessor that changes a number on this path, :105-154)
    qz = np.rint(synth.fractal(72, 88, seed=17, top_shift=5, n_octaves=5, zrange=90.0)).astype(np.float64)
    base = dict(fill_flats=False, drain_pits_path=False)
    for nm, kw in (('minborder', dict(drain_pits_min_border=True)),
                   ('shortreach', dict(drain_pits_max_iter=5, drain_pits_max_dist=3)),
                   ('xyreach', dict(drain_pits_max_dist_XY=70.0)),
                   ('ucalimit', dict(apply_uca_limit_edges=True, uca_saturation_limit=2.0)),
                   ('twilimits', dict(apply_twi_limits=True, apply_twi_limits_on_uca=True, twi_min_slope=0.01, uca_saturation_limit=4.0))):
        k2 = dict(base); k2.update(kw)
        cases.append(('g5_opt_' + nm, qz, 10.0, 12.0, k2))
    qi = np.rint(synth.fractal(64, 72, seed=19, top_shift=5, n_octaves=5, zrange=50.0) - 8.0).astype(np.int16)
    cases.append(('g5_opt_cond_a', qi, 30.0, 30.0, dict(maximum_pit_area=4, fill_flats_source_tol=0, fill_flats_peaks=False)))
    cases.append(('g5_opt_cond_b', qi, 30.0, 30.0, dict(fill_flats_pits=False, fill_flats_below_sea=True, drain_pits_max_iter=8,
                                                        drain_pits_max_dist=4)))
    # float32 DEMs (what most GeoTIFF DEMs are): with the conditioning off numpy subtracts elevations in float32 and only
    # the division by the float64 spacing promotes (:1958-1962, :1361); drain_pits_path edits the float32 surface in place
    # (:543-544); fill_flats converts to float64 first (:561)
    # (low, rough relief around zero: neighbours differ by more than a factor two, where a float32 difference is inexact)
    f32 = (synth.fractal(80, 104, seed=23, top_shift=3, n_octaves=4, zmin=-1.5, zrange=37.7) + 0.123456789).astype(np.float32)
    cases.append(('g5_f32_pits', f32, 29.7, 31.3, dict(fill_flats=False, drain_pits_path=False)))
    cases.append(('g5_f32_nopits', f32, 29.7, 31.3, dict(fill_flats=False, drain_pits=False, drain_pits_path=False)))
    cases.append(('g5_f32_pathonly', f32, 29.7, 31.3, dict(fill_flats=False)))
    cases.append(('g5_f32_defaults', f32, 29.7, 31.3, dict()))
    i32 = np.rint(synth.fractal(64, 80, seed=29, top_shift=5, n_octaves=5, zrange=70000.0)).astype(np.int32)
    cases.append(('g5_int32_pits', i32, 30.0, 30.0, dict(fill_flats=False, drain_pits_path=False)))
    # degenerate tiles: the smallest legal tile, one-cell-wide interiors, constant / all-sea / all-nodata tiles, a
    # checkerboard (every other cell a pit), one pit in a bowl, negative and very large elevations
    rng = np.random.default_rng(3)
    deg = {
        'ramp3x3': np.arange(9, dtype=float).reshape(3, 3) + 1,
        'ramp3x7': (np.arange(21, dtype=float).reshape(3, 7) % 5) + 1,
        'ramp7x3': (np.arange(21, dtype=float).reshape(7, 3) % 4) + 1,
        'const8': np.full((8, 8), 5.0),
        'zeros6': np.zeros((6, 6)),
        'allnan6': np.full((6, 6), np.nan),
        'checker10': (np.indices((10, 10)).sum(0) % 2).astype(float) * 3 + 2,
        'onepit9': np.maximum(np.abs(np.arange(9) - 4)[:, None], np.abs(np.arange(9) - 4)[None, :]).astype(float) + 1,
        'neg8': -(rng.random((8, 8)) * 10 + 1),
        'huge8': rng.random((8, 8)) * 1e12 + 1e13,
    }
    for nm, z in deg.items():
        cases.append(('g5_deg_%s_pits' % nm, z, 2.0, 3.0, dict(fill_flats=False, drain_pits_path=False)))
        cases.append(('g5_deg_%s_nopits' % nm, z, 2.0, 3.0, dict(fill_flats=False, drain_pits=False, drain_pits_path=False)))
    if '--only-nan' in sys.argv:
        cases = [c for c in cases if 'nan' in c[0] or 'sea' in c[0] or '_opt_' in c[0]]
    if '--only-deg' in sys.argv:
        cases = [c for c in cases if '_deg_' in c[0]]
    if '--only-f32' in sys.argv:
        cases = [c for c in cases if '_f32_' in c[0] or '_int32_' in c[0]]

    for name, elev, dX, dY, kw in cases:
        rec = run_case(elev, dX, dY, **kw)
        save(name, rec, kw)
    if '--only-nan' in sys.argv or '--only-f32' in sys.argv or '--only-deg' in sys.argv:
        write_manifest()
        return

    edge_cases()
    conditioning_cases()
    write_manifest()


def edge_cases():
    fr = synth.fractal(96, 128, seed=0, top_shift=6, n_octaves=6)
    run_edge_update_case('g6_edge_update_nopits', fr, 30.0, 30.0, 1, drain_pits=False)
    run_edge_update_case('g6_edge_update_pits', fr, 30.0, 30.0, 2, drain_pits=True)
    run_edge_update_case('g6_edge_update_cone', synth.cone_scaled(48), 1.0, 1.0, 3, drain_pits=True)
    quant = np.rint(synth.fractal(64, 64, seed=9, top_shift=5, n_octaves=5, zrange=60.0)).astype(np.float64)
    run_edge_update_case('g6_edge_update_quant', quant, 10.0, 10.0, 4, drain_pits=True)


def write_manifest():
    # manifest with sha256 of every fixture
    lines = []
    for fn in sorted(os.listdir(OUT)):
        if fn.endswith('.npz'):
            lines.append('%s  %s' % (hashlib.sha256(open(os.path.join(OUT, fn), 'rb').read()).hexdigest(), fn))
    open(os.path.join(OUT, 'SHA256SUMS'), 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
