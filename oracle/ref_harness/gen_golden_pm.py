"""Directory-flow goldens: run the UNMODIFIED reference ProcessManager (n_workers=1, DEBUG spacing
as in its own tests, pydem/test/test_end_to_end.py:62) on small multi-tile mosaics and capture,
per tile, every array it stores plus the grid bookkeeping our ProcessManager restates.

    oracle/ref_harness/run.sh oracle/ref_harness/gen_golden_pm.py

Fixtures (tests/golden/pm_*.npz) hold data only.  Build container only.
"""
import glob
import os
import shutil
import sys
import tempfile
import warnings

from load_reference import load_reference
pydem = load_reference()

import numpy as np  # noqa: E402
from pydem import process_manager, utils  # noqa: E402

REPO = os.path.abspath(os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, REPO)
from pydem_amd import synth  # noqa: E402
from gen_golden import OUT, write_manifest  # noqa: E402


def chunk_edges(NN, n_chunks, overlap):
    """Tile boundaries used for the mosaics (same construction as the reference's test helper
    mk_test_multifile, utils_test_pydem.py:371-379)."""
    size = int(np.ceil(NN / n_chunks))
    lo = np.arange(0, NN - overlap, size)
    lo[1:] -= overlap // 2
    hi = np.arange(0, NN - overlap, size)
    hi[:-1] = hi[1:] + int(np.ceil(overlap / 2))
    hi[-1] = NN
    return lo, np.minimum(hi, NN)


def run_pm_case(name, raster, ny_grid, nx_grid, overlap, dp_kwargs):
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, 'chunks')
    os.makedirs(path)
    ni, nj = raster.shape
    lat = np.linspace(46, 45, ni)
    lon = np.linspace(-73, -72, nj)
    te_, be_ = chunk_edges(ni, ny_grid, overlap)
    le_, re_ = chunk_edges(nj, nx_grid, overlap)
    for te, be in zip(te_, be_):
        for le, re in zip(le_, re_):
            fn = os.path.join(path, utils.get_fn_from_coords((lat[be - 1], lon[le], lat[te], lon[re - 1]), 'elev'))
            utils.mk_geotiff_obj(raster[te:be, le:re], fn, bands=1, lat=[lat[te], lat[be - 1]], lon=[lon[le], lon[re - 1]])
    process_manager.DEBUG = True
    pm = process_manager.ProcessManager(in_path=path, n_workers=1, _debug=False, dem_proc_kwargs=dict(dp_kwargs))
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        pm.process_twi()
        pm.save_non_overlap_data()
    rec = {'raster': raster, 'n_tiles': np.int64(pm.n_inputs), 'index': np.array(pm.index),
           'grid_id': np.array(pm.grid_id), 'grid_id2i': np.array(pm.grid_id2i),
           'grid_size_tot': np.array(pm.grid_size_tot), 'grid_size_tot_unique': np.array(pm.grid_size_tot_unique)}
    s2a = lambda s: [s.start, s.stop] if isinstance(s, slice) else [int(s), int(s) + 1]
    for i in range(pm.n_inputs):
        slc = pm.grid_slice[i]
        tile = utils.read_raster(pm.elev_source_files[i])
        rec['t%02d_in_elev' % i] = tile.read(1)
        rec['t%02d_bounds' % i] = np.array(tile.bounds)
        for key in ('elev', 'aspect', 'slope', 'uca', 'uca_edges', 'edge_todo', 'edge_done', 'twi'):
            rec['t%02d_%s' % (i, key)] = np.array(pm.out_file[key][slc])
        rec['t%02d_grid_slice' % i] = np.array([s2a(slc[0]), s2a(slc[1])])
        rec['t%02d_grid_slice_unique' % i] = np.array([s2a(pm.grid_slice_unique[i][0]), s2a(pm.grid_slice_unique[i][1])])
        rec['t%02d_grid_slice_noverlap' % i] = np.array([s2a(pm.grid_slice_noverlap[i][0]), s2a(pm.grid_slice_noverlap[i][1])])
        ed = pm.edge_data[i]
        rec['t%02d_edge_data' % i] = np.array([[s2a(ed[k][0]), s2a(ed[k][1])] for k in
                                               ('left', 'right', 'top', 'bottom', 'top-left', 'top-right',
                                                'bottom-left', 'bottom-right')])
    for key in ('elev', 'uca', 'aspect', 'slope', 'twi'):
        rec['compact_' + key] = np.array(pm.out_file_noverlap[key][:])
    rec['kwargs_repr'] = np.array(repr(sorted(dict(dp_kwargs, ny_grid=ny_grid, nx_grid=nx_grid, overlap=overlap).items())))
    if name is None:                       # in-memory use (soak_pm_reference.py)
        shutil.rmtree(tmp)
        return rec
    os.makedirs(OUT, exist_ok=True)
    fn = os.path.join(OUT, name + '.npz')
    np.savez_compressed(fn, **rec)
    print('%-30s tiles=%d  %.1f KiB' % (name, pm.n_inputs, os.path.getsize(fn) / 1024.))
    shutil.rmtree(tmp)


def main():
    cone = synth.cone_scaled(32)          # the input of the reference's own multi-tile test (case 33, NN=32)
    nopath = dict(drain_pits_path=False)
    if '--twilimits-only' in sys.argv:     # (added later: writes this one fixture and the manifest, leaves the others alone)
        fr = synth.fractal(60, 72, seed=17, top_shift=5, n_octaves=5, zrange=200.0)
        # the TWI limits in the directory flow: the calc_twi worker builds a fresh processor (twi_min_area = inf unless given)
        run_pm_case('pm_fractal_twilimits_2x2_ov2', fr, 2, 2, 2, dict(drain_pits_path=False, apply_twi_limits=True, apply_twi_limits_on_uca=True,
                                                                       twi_min_slope=0.01, uca_saturation_limit=4.0))
        write_manifest()
        return
    run_pm_case('pm_cone32_3x3_ov2', cone, 3, 3, 2, {})
    run_pm_case('pm_cone32_3x3_ov1', cone, 3, 3, 1, {})
    run_pm_case('pm_cone32_4x5_ov3', cone, 4, 5, 3, {})
    fr = synth.fractal(60, 72, seed=17, top_shift=5, n_octaves=5, zrange=200.0)
    run_pm_case('pm_fractal_2x2_ov2', fr, 2, 2, 2, nopath)
    run_pm_case('pm_fractal_2x3_ov1', fr, 2, 3, 1, nopath)
    run_pm_case('pm_fractal_2x2_ov1_nopits', fr, 2, 2, 1, dict(drain_pits_path=False, drain_pits=False))
    if '--base-only' in sys.argv:
        write_manifest()
        return
    # nodata and sea level across tile edges (NaN strips in the edge exchange, `elev > 0` gates)
    frn = synth.fractal(64, 80, seed=31, top_shift=5, n_octaves=5, zrange=300.0) - 60.0
    frn[frn < 0] = 0.0
    frn[24:40, 30:46] = np.nan          # a nodata block that straddles the 2x2 tile cross
    frn[0:6, 60:] = np.nan
    run_pm_case('pm_nansea_2x2_ov2', frn, 2, 2, 2, nopath)
    # float32 tiles, conditioning off: the tiles subtract elevations in float32 (dem_processing.py:1958-1962)
    f32 = (synth.fractal(56, 64, seed=37, top_shift=3, n_octaves=4, zmin=-1.0, zrange=33.3) + 0.123456789).astype(np.float32)
    run_pm_case('pm_f32_2x2_ov1', f32, 2, 2, 1, dict(fill_flats=False, drain_pits_path=False))
    # quantised int16 tiles with the reference's default options (conditioning inside every tile)
    q16 = np.rint(synth.fractal(48, 60, seed=41, top_shift=5, n_octaves=5, zrange=80.0)).astype(np.int16)
    run_pm_case('pm_int16_defaults_2x2_ov2', q16, 2, 2, 2, {})
    write_manifest()


if __name__ == '__main__':
    main()
