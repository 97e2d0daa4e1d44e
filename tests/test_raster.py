"""pydem_amd/raster.py: the GeoTIFF reader against the reference's own test raster and against files built here with
every layout it claims to read; Vincenty against published lines; the spacing rules of the reference; and the
directory flow fed with GeoTIFF tiles instead of .npz tiles (CPU only)."""
import os
import struct
import zlib

import numpy as np
import pytest

from conftest import load_golden
from pydem_amd import raster

HERE = os.path.dirname(os.path.abspath(__file__))


def test_reads_the_reference_test_raster():
    """pydem/test/test_NN032_033_elev.tif (a data file of the reference's tests): same samples as the golden captured
    when the reference ran on it, and the georeferencing its file name spells (N46.0/W73.0 ... N45.0/W72.0, 32 px)."""
    ds = raster.read_geotiff(os.path.join(HERE, 'golden', 'ref_test_NN032_033_elev.tif'))
    g = load_golden('g3_tif32')
    assert ds.array.dtype == np.float64 and np.array_equal(ds.array, g['in_elev'])
    assert not ds.is_projected and ds.ellipsoid == 'WGS-84'
    a, b, c, d, e, f = ds.transform
    assert abs(a - 1 / 31.0) < 1e-12 and abs(e + 1 / 31.0) < 1e-12 and b == 0 and d == 0
    assert np.allclose(ds.bounds, (-73 - 0.5 / 31, 45 - 0.5 / 31, -72 + 0.5 / 31, 46 + 0.5 / 31))


def _lzw_encode(data):
    """Textbook TIFF LZW encoder (test-side only) to exercise the decoder."""
    codes, table, nbits = [], {bytes([i]): i for i in range(256)}, 9
    out, acc, have = bytearray(), 0, 0

    def emit(code, nb):
        nonlocal acc, have
        acc = (acc << nb) | code; have += nb
        while have >= 8:
            out.append((acc >> (have - 8)) & 0xFF); have -= 8

    emit(256, 9)
    nxt, w = 258, b''
    for byte in data:
        wc = w + bytes([byte])
        if wc in table:
            w = wc
            continue
        emit(table[w], nbits)
        table[wc] = nxt; nxt += 1
        if nxt == (1 << nbits) - 1 + 1 and nbits < 12:      # early change: widen one code early
            nbits += 1
        if nxt == 4094:
            emit(256, nbits)
            table, nbits, nxt = {bytes([i]): i for i in range(256)}, 9, 258
        w = bytes([byte])
    if w:
        emit(table[w], nbits)
    emit(257, nbits)
    if have:
        out.append((acc << (8 - have)) & 0xFF)
    return bytes(out)


def _packbits_encode(data):
    out, i = bytearray(), 0
    while i < len(data):
        n = min(128, len(data) - i)
        out.append(n - 1); out += data[i:i + n]; i += n
    return bytes(out)


def _build_tiff(arr, bo='<', comp=1, pred=1, tile=None, rows_per_strip=None, big=False):
    """Single-band TIFF with the requested byte order / compression / predictor / layout (test-side writer)."""
    a = np.ascontiguousarray(arr).astype(arr.dtype.newbyteorder(bo))
    h, w = a.shape
    kind = {'u': 1, 'i': 2, 'f': 3}[a.dtype.kind]

    def pack_block(blk):
        b = np.ascontiguousarray(blk)
        if pred == 2:
            nat = b.astype(b.dtype.newbyteorder('='))
            d = nat.copy(); d[:, 1:] = nat[:, 1:] - nat[:, :-1]
            b = d.astype(a.dtype)
        raw = b.tobytes()
        return {1: lambda r: r, 5: _lzw_encode, 8: lambda r: zlib.compress(r), 32773: _packbits_encode}[comp](raw)

    chunks = []
    if tile:
        th, tw = tile
        for r0 in range(0, h, th):
            for c0 in range(0, w, tw):
                blk = np.zeros((th, tw), a.dtype)
                sub = a[r0:r0 + th, c0:c0 + tw]
                blk[:sub.shape[0], :sub.shape[1]] = sub
                chunks.append(pack_block(blk))
    else:
        rps = rows_per_strip or h
        for r0 in range(0, h, rps):
            chunks.append(pack_block(a[r0:r0 + rps]))
    off_fmt, hdr = ('Q', 16) if big else ('I', 8)
    entries = [(256, 4, [w]), (257, 4, [h]), (258, 3, [a.dtype.itemsize * 8]), (259, 3, [comp]), (262, 3, [1]), (277, 3, [1]),
               (317, 3, [pred]), (339, 3, [kind]), (33550, 12, [0.5, 0.25, 0.0]), (33922, 12, [0.0, 0.0, 0.0, 100.0, 200.0, 0.0]),
               (34735, 3, [1, 1, 0, 1, 1024, 0, 1, 1]), (42113, 2, b'-9999\x00')]
    if tile:
        entries += [(322, 4, [tile[1]]), (323, 4, [tile[0]]), (324, 16 if big else 4, None), (325, 16 if big else 4, [len(c) for c in chunks])]
    else:
        entries += [(278, 4, [rows_per_strip or h]), (273, 16 if big else 4, None), (279, 16 if big else 4, [len(c) for c in chunks])]
    entries.sort(key=lambda t: t[0])
    esz, inline = (20, 8) if big else (12, 4)
    ifd_len = (8 if big else 2) + esz * len(entries) + (8 if big else 4)
    extra_off = hdr + ifd_len
    fmts = {2: 'c', 3: 'H', 4: 'I', 12: 'd', 16: 'Q'}
    sizes = {2: 1, 3: 2, 4: 4, 12: 8, 16: 8}
    # first pass: sizes of out-of-line values, then the data offsets
    extra_len = 0
    for tag, typ, vals in entries:
        cnt = len(chunks) if vals is None else len(vals)
        sz = sizes[typ] * cnt
        if sz > inline:
            extra_len += sz + (sz % 2)
    data_off = extra_off + extra_len
    offs, o = [], data_off
    for c in chunks:
        offs.append(o); o += len(c)
    body, extra = b'', b''
    for tag, typ, vals in entries:
        if vals is None:
            vals = offs
        raw = vals if typ == 2 else struct.pack(bo + fmts[typ] * len(vals), *vals)
        cnt = len(vals)
        if len(raw) <= inline:
            val = raw + b'\x00' * (inline - len(raw))
        else:
            val = struct.pack(bo + off_fmt, extra_off + len(extra))
            extra += raw + (b'\x00' if len(raw) % 2 else b'')
        body += struct.pack(bo + 'HH' + ('Q' if big else 'I'), tag, typ, cnt) + val
    head = (b'II' if bo == '<' else b'MM')
    if big:
        head += struct.pack(bo + 'HHHQ', 43, 8, 0, 16)
        ifd = struct.pack(bo + 'Q', len(entries)) + body + struct.pack(bo + 'Q', 0)
    else:
        head += struct.pack(bo + 'HI', 42, 8)
        ifd = struct.pack(bo + 'H', len(entries)) + body + struct.pack(bo + 'I', 0)
    return head + ifd + extra + b''.join(chunks)


@pytest.mark.parametrize('dtype', ['int16', 'uint16', 'int32', 'float32', 'float64', 'uint8'])
@pytest.mark.parametrize('layout', [dict(), dict(comp=8), dict(comp=5), dict(comp=32773), dict(comp=5, pred=2), dict(comp=8, pred=2, tile=(16, 32)),
                                    dict(bo='>'), dict(bo='>', comp=5, tile=(32, 16)), dict(rows_per_strip=7, comp=8), dict(big=True, comp=8),
                                    dict(big=True, bo='>', tile=(16, 16))])
def test_reads_every_layout(dtype, layout, tmp_path):
    if layout.get('pred') == 2 and dtype.startswith('float'):
        pytest.skip("horizontal differencing is defined for integer samples")
    rng = np.random.default_rng(7)
    base = rng.integers(0, 200, (45, 70))
    base = np.cumsum(base, axis=1) % 250 if dtype == 'uint8' else base * 13 - 900
    arr = base.astype(dtype)
    if dtype.startswith('float'):
        arr = arr + rng.random((45, 70)).astype(dtype)
    fn = str(tmp_path / 't.tif')
    open(fn, 'wb').write(_build_tiff(arr, **layout))
    ds = raster.read_geotiff(fn)
    assert ds.array.dtype == np.dtype(dtype) and np.array_equal(ds.array, arr)
    assert ds.is_projected and ds.nodata == -9999.0
    assert ds.transform == (0.5, 0.0, 100.0, 0.0, -0.25, 200.0)
    assert ds.bounds == (100.0, 200.0 - 0.25 * 45, 100.0 + 0.5 * 70, 200.0)


@pytest.mark.parametrize('compress', [False, True])
def test_writer_round_trip(compress, tmp_path):
    arr = (np.arange(37 * 53).reshape(37, 53) % 311).astype('int16')
    fn = str(tmp_path / 'w.tif')
    raster.write_geotiff(fn, arr, (1 / 3600.0, 0.0, -72.5, 0.0, -1 / 3600.0, 45.25), projected=False, nodata=-32768, compress=compress)
    ds = raster.read_geotiff(fn)
    assert np.array_equal(ds.array, arr) and not ds.is_projected and ds.nodata == -32768.0
    assert np.allclose(ds.transform, (1 / 3600.0, 0.0, -72.5, 0.0, -1 / 3600.0, 45.25), rtol=0, atol=1e-15)


def test_vincenty_published_lines():
    # Geoscience Australia's GDA94 test line Flinders Peak -> Buninyong (GRS80): 54 972.271 m
    d = raster.geodesic_m(-(37 + 57 / 60 + 3.72030 / 3600), 144 + 25 / 60 + 29.52440 / 3600,
                          -(37 + 39 / 60 + 10.15610 / 3600), 143 + 55 / 60 + 35.38390 / 3600, 'GRS-80')
    assert abs(d - 54972.271) < 1e-3
    assert abs(raster.geodesic_m(0, 0, 90, 0) - 10001965.7293) < 1e-3           # WGS-84 quarter meridian
    assert abs(raster.geodesic_m(0, 10, 0, 11) - 6378137.0 * np.pi / 180) < 1e-6  # one degree of the equator
    assert raster.geodesic_m(12.5, 7.0, 12.5, 7.0) == 0.0
    assert abs(raster.geodesic_m(45, 0, 45, 1 / 3600) - raster.geodesic_m(45, 5, 45, 5 + 1 / 3600)) < 1e-9


def test_spacing_rules_of_the_reference():
    n = 6
    t = (1 / 3600.0, 0.0, -72.0, 0.0, -1 / 3600.0, 45.0)
    dX, dY, dX2, dY2 = raster.spacing_from_geotransform(n, t, False)
    assert dX.shape == (n - 1,) and dY.shape == (n - 1,) and dX2.shape == (n,) and dY2.shape == (n,)
    lat0 = 45.0 - 0.5 / 3600                                                     # transform.f + dy / 2 (utils.py:155)
    for j in range(n - 1):
        assert abs(dX[j] - raster.geodesic_m(lat0 - (j + 1) / 3600, 0.0, lat0 - (j + 1) / 3600, 1 / 3600)) < 1e-9   # longitude anchor irrelevant
        assert abs(dY[j] - raster.geodesic_m(lat0 - j / 3600, 0.0, lat0 - (j + 1) / 3600, 0.0)) < 1e-9
    assert np.all(np.diff(dX) > 0)                      # going south from 45 N the parallels get longer
    assert 30.8 < dY.mean() < 30.9 and 21.8 < dX.mean() < 22.0
    p = raster.spacing_from_geotransform(n, (30.0, 0, 5e5, 0, -25.0, 4e6), True)
    assert np.array_equal(p[0], np.full(n - 1, 30.0)) and np.array_equal(p[1], np.full(n - 1, 25.0))
    assert np.array_equal(p[2], np.full(n, 30.0)) and np.array_equal(p[3], np.full(n, 25.0))


def test_directory_flow_from_geotiff_tiles(tmp_path):
    """The same mosaic as .npz tiles and as (projected, Deflate) GeoTIFF tiles must give the same results."""
    from oracle_processor import OracleProcessor
    from pydem_amd import process_manager
    from test_process_manager_cpu import run_pm
    g = load_golden('pm_cone32_3x3_ov2')
    pm0, compact0, order0 = run_pm(g, str(tmp_path / 'npz'), processor_cls=OracleProcessor)
    tif_dir = tmp_path / 'tif'
    os.makedirs(tif_dir)
    for i in range(int(g['n_tiles'])):
        elev = g['t%02d_elev' % i]
        left, bottom, right, top = [float(v) for v in g['t%02d_bounds' % i]]
        n, m = elev.shape
        raster.write_geotiff(str(tif_dir / ('tile_%03d.tif' % i)), elev, ((right - left) / m, 0.0, left, 0.0, -(top - bottom) / n, top),
                             projected=True, compress=True)
    dkw = {k: v for k, v in g['kwargs'].items() if k not in ('ny_grid', 'nx_grid', 'overlap')}
    process_manager.DEBUG = True
    try:
        pm1 = process_manager.ProcessManager(in_path=str(tif_dir), dem_proc_kwargs=dkw, elev_conditioned=True, processor_cls=OracleProcessor)
        import warnings
        with warnings.catch_warnings():
            warnings.simplefilter('ignore')
            pm1.process_twi()
            compact1 = pm1.save_non_overlap_data()
    finally:
        process_manager.DEBUG = False
    for key in ('elev', 'uca', 'aspect', 'slope', 'twi'):
        assert np.array_equal(compact0[key], compact1[key], equal_nan=True), key
    # export (reference save_geotiff :862-931) and read back
    out = str(tmp_path / 'uca.tif')
    pm1.save_geotiff(out, 'uca', 'float32')
    ds = raster.read_geotiff(out)
    assert ds.array.dtype == np.float32 and np.array_equal(ds.array, compact1['uca'].astype('float32'), equal_nan=True)
    assert ds.is_projected and abs(ds.bounds[0] - pm1.index[:, 0].min()) < 1e-9 and abs(ds.bounds[3] - pm1.index[:, 3].max()) < 1e-9
    # ... with 'average' overviews (:927-931): further images in the file, same ground, block means of the full raster
    out2 = str(tmp_path / 'uca_ov.tif')
    pm1.save_geotiff(out2, 'uca', 'float64', overview_type='average', overview_factors=[3, 9])
    full = raster.read_geotiff(out2)
    ov1, ov2 = raster.read_geotiff(out2, 1), raster.read_geotiff(out2, 2)
    assert np.array_equal(full.array, compact1['uca'], equal_nan=True)
    want1 = raster.block_mean_overview(compact1['uca'], 3, like_reference=False)
    assert np.array_equal(ov1.array, want1, equal_nan=True) and ov1.shape == tuple(-(-n // 3) for n in full.shape)
    assert np.array_equal(ov2.array, raster.block_mean_overview(want1, 3, like_reference=False), equal_nan=True)
    assert np.allclose(ov1.bounds, full.bounds) and np.allclose(ov2.bounds, full.bounds)
    with pytest.raises(IndexError):
        raster.read_geotiff(out2, 3)
    with pytest.raises(NotImplementedError):
        pm1.save_geotiff(out2, 'uca', 'float64', overview_type='cubic')
    # the reference's file layout (:906-913: 512 x 512 blocks, BigTIFF) is the default; other resampling kinds; rescale tag (:925)
    with open(out2, 'rb') as fh:
        assert struct_magic(fh.read(4)) == 43
    out3 = str(tmp_path / 'uca_max.tif')
    pm1.save_geotiff(out3, 'uca', 'int32', rescale=(0.0, 10.0, 100.0), overview_type='max', overview_factors=[3], blocksize=16,
                     bigtiff=False, nodata=-1)
    scaled = (compact1['uca'] - 0.0) / 10.0 * 100.0
    with np.errstate(invalid='ignore'):
        assert np.array_equal(raster.read_geotiff(out3).array, scaled.astype('int32'))
        ovm = raster.block_overview(scaled, 3, 'max')
        assert np.array_equal(raster.read_geotiff(out3, 1).array, np.where(np.isnan(ovm), -1, ovm).astype('int32'))
    with open(out3, 'rb') as fh:
        raw = fh.read()
    assert struct_magic(raw[:4]) == 42 and b'<Item name="rescale">0.0,10.0,100.0</Item>' in raw and b'rio_overview_resampling">max<' in raw
    # one file per key (save_non_overlap_data_geotiff :786-860)
    pm1.save_non_overlap_data_geotiff('float32', new_path=str(tmp_path / 'tiffs'), keys=('elev', 'twi'), overview_type='average')
    for key in ('elev', 'twi'):
        ds2 = raster.read_geotiff(str(tmp_path / 'tiffs' / (key + '.tiff')))
        assert np.array_equal(ds2.array, compact1[key].astype('float32'), equal_nan=True), key
        assert raster.read_geotiff(str(tmp_path / 'tiffs' / (key + '.tiff')), 1).shape == tuple(-(-n // 3) for n in ds2.shape)
    # the overview pyramid of the stitched arrays (process_overviews :933-991)
    pyr = pm1.process_overviews(out_path=str(tmp_path / 'ov'), keys=('uca', 'twi'), overviews=(3, 9, 27, 81))
    lvl = np.asarray(compact1['uca'], np.float64)
    names = []
    for ov in (3, 9, 27, 81):
        if any(-(-n // 3) <= 3 for n in lvl.shape):
            break
        lvl = raster.block_mean_overview(lvl, 3)
        names.append('uca_%d' % ov)
        assert np.array_equal(pyr['uca_%d' % ov], lvl, equal_nan=True)
        assert np.array_equal(np.load(str(tmp_path / 'ov' / ('uca_%d.npy' % ov))), lvl, equal_nan=True)
    assert names and sorted(k for k in pyr if k.startswith('uca_')) == sorted(names)


def test_block_mean_overview_equals_the_reference_function():
    """raster.block_mean_overview against calc_overview of the unmodified reference (pydem/process_manager.py:317-352;
    tests/golden/overview_cases.npz, written by oracle/ref_harness/gen_golden_overview.py): full blocks, partial blocks on
    the right / bottom / corner (including the reference's reshape of the bottom row), NaN cells, an all-zero array --
    bit for bit.  With like_reference=False the bottom row holds the plain partial-block means."""
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'overview_cases.npz'))
    for k in range(int(g['n_cases'])):
        out = raster.block_mean_overview(g['in_%d' % k], int(g['factor_%d' % k]))
        assert out.shape == g['out_%d' % k].shape and np.array_equal(out, g['out_%d' % k], equal_nan=True), k
    z = np.arange(35, dtype=float).reshape(5, 7)
    plain = raster.block_mean_overview(z, 2, like_reference=False)
    assert plain.shape == (3, 4)
    assert plain[2, 1] == z[4, 2:4].mean() and plain[0, 3] == z[0:2, 6].mean() and plain[2, 3] == z[4, 6] and plain[1, 1] == z[2:4, 2:4].mean()


def test_dem_processor_elev_fn_constructor(monkeypatch):
    """DEMProcessor(elev_fn=...) (reference dem_processing.py:229-232 -> utils.dem_processor_from_raster_kwargs :46-51): the
    raster's array, per-row spacing, bounds and transform become constructor arguments; explicit keywords win.  (No device
    call here: the constructor only stores host arrays.)"""
    from pydem_amd import DEMProcessor
    fn = os.path.join(HERE, 'golden', 'ref_test_NN032_033_elev.tif')
    ds = raster.read_geotiff(fn)
    dp = DEMProcessor(elev_fn=fn)
    assert np.array_equal(np.asarray(dp.elev), np.asarray(ds.read(1)))
    want = raster.dem_processor_from_raster_kwargs(fn)
    for k in ('dX', 'dY', 'dX2', 'dY2'):
        assert np.array_equal(getattr(dp, k), want[k]), k
    assert dp.dX.shape == (ds.shape[0] - 1,) and dp.dX2.shape == (ds.shape[0],)
    assert list(dp.bounds) == list(ds.bounds) and list(dp.transform) == list(ds.transform)
    dp2 = DEMProcessor(elev_fn=fn, dX=2.0, fill_flats=False)
    assert np.array_equal(dp2.dX, np.full(ds.shape[0] - 1, 2.0)) and dp2.fill_flats is False
    assert np.array_equal(dp2.dY, want['dY'])


def test_integer_overviews_carry_nodata_not_nan(tmp_path):
    """Block-mean overviews of an integer raster: a block without valid cells is NaN in the float mean; in the file it must be
    the nodata value (announced by the GDAL_NODATA tag), never an undefined integer -- and without a nodata value the export
    is refused."""
    import warnings
    base = np.arange(36, dtype=np.int16).reshape(6, 6)
    ov = np.array([[1.5, np.nan], [20.25, 30.0]])
    out = str(tmp_path / 'i16.tif')
    with warnings.catch_warnings():
        warnings.simplefilter('error')                          # (the NaN -> int cast used to raise a RuntimeWarning)
        raster.write_geotiff(out, base, (1.0, 0.0, 0.0, 0.0, -1.0, 6.0), projected=True, nodata=-32768, overviews=[ov])
    got = raster.read_geotiff(out, 1)
    assert got.array.dtype == np.int16 and got.nodata == -32768
    assert got.array.tolist() == [[1, -32768], [20, 30]]
    with pytest.raises(ValueError):
        raster.write_geotiff(out, base, (1.0, 0.0, 0.0, 0.0, -1.0, 6.0), projected=True, overviews=[ov])


def test_geodesic_known_answers():
    """Published known answers for the geodesic that replaces geopy.distance in the spacing rules (pydem/utils.py:127-174 calls
    geopy, which is not available here; the harness cannot capture its output):
      * Vincenty's own check line Flinders Peak -> Buninyong on GRS-80 (Geoscience Australia's worked example): 54 972.271 m;
      * the WGS-84 quarter meridian (equator to pole): 10 001 965.729 m;
      * one degree of longitude on the equator: a * pi / 180."""
    d = raster.geodesic_m(-(37 + 57 / 60 + 3.72030 / 3600), 144 + 25 / 60 + 29.52440 / 3600,
                          -(37 + 39 / 60 + 10.15610 / 3600), 143 + 55 / 60 + 35.38390 / 3600, 'GRS-80')
    assert abs(d - 54972.271) < 1e-3
    assert abs(raster.geodesic_m(0.0, 10.0, 90.0, 10.0) - 10001965.729) < 1e-3
    assert abs(raster.geodesic_m(0.0, 0.0, 0.0, 1.0) - 6378137.0 * np.pi / 180) < 1e-6
    # symmetry and additivity along a meridian (what dY of consecutive rows relies on)
    a, b, c = raster.geodesic_m(40.0, 7.0, 40.5, 7.0), raster.geodesic_m(40.5, 7.0, 41.0, 7.0), raster.geodesic_m(40.0, 7.0, 41.0, 7.0)
    assert abs(a + b - c) < 1e-6 and raster.geodesic_m(41.0, 7.0, 40.0, 7.0) == c


@pytest.mark.parametrize('dtype', ['float64', 'float32', 'int16', 'uint8'])
@pytest.mark.parametrize('bigtiff', [False, True])
def test_tiled_and_bigtiff_round_trip(dtype, bigtiff, tmp_path):
    """The layout of the reference's export (512 x 512 blocks in a BigTIFF, pydem/process_manager.py:906-913): tiles whose
    edge blocks stick out of the raster, 64-bit offsets, overviews as further directories, metadata items -- written by
    write_geotiff and read back by read_geotiff (which also reads what GDAL writes: tests above)."""
    rng = np.random.default_rng(5)
    arr = (rng.random((150, 233)) * 200).astype(dtype)
    ovs = [raster.block_overview(arr, 3, 'nearest'), raster.block_overview(arr, 9, 'nearest')]
    out = str(tmp_path / 't.tif')
    raster.write_geotiff(out, arr, (0.5, 0.0, 10.0, 0.0, -0.5, 99.0), projected=True, compress=True, overviews=ovs, tile=64,
                         bigtiff=bigtiff, nodata=(None if dtype.startswith('f') else 0), tags={'rescale': '0,1,2'})
    with open(out, 'rb') as fh:
        assert struct_magic(fh.read(4)) == (43 if bigtiff else 42)
    ds = raster.read_geotiff(out)
    assert ds.array.dtype == np.dtype(dtype) and np.array_equal(ds.array, arr)
    assert ds.transform == (0.5, 0.0, 10.0, 0.0, -0.5, 99.0) and ds.is_projected
    for k, o in enumerate(ovs):
        got = raster.read_geotiff(out, k + 1)
        assert np.array_equal(got.array, o.astype(dtype)) and got.shape == o.shape
    with pytest.raises(IndexError):
        raster.read_geotiff(out, 3)
    with pytest.raises(ValueError):
        raster.write_geotiff(out, arr, (0.5, 0.0, 10.0, 0.0, -0.5, 99.0), tile=100)


def struct_magic(b):
    import struct
    assert b[:2] == b'II'
    return struct.unpack('<H', b[2:4])[0]


def test_block_overview_kinds_against_loops():
    """Every overview resampling kind against a plain loop over the blocks (partial edge blocks, NaN = no data ignored)."""
    rng = np.random.default_rng(11)
    a = np.round(rng.random((14, 17)) * 6)
    a[rng.random(a.shape) < 0.15] = np.nan
    a[:3, :3] = np.nan                                     # one block without data
    f = 3
    N, M = -(-a.shape[0] // f), -(-a.shape[1] // f)
    import warnings

    def loop(fn):
        out = np.full((N, M), np.nan)
        for i in range(N):
            for j in range(M):
                v = a[i * f:(i + 1) * f, j * f:(j + 1) * f].ravel()
                v = v[~np.isnan(v)]
                if v.size:
                    out[i, j] = fn(v)
        return out

    def mode(v):
        vals, cnt = np.unique(v, return_counts=True)
        return vals[np.argmax(cnt)]
    want = {'max': np.max, 'min': np.min, 'med': np.median, 'q1': lambda v: np.quantile(v, 0.25), 'q3': lambda v: np.quantile(v, 0.75),
            'sum': np.sum, 'rms': lambda v: np.sqrt(np.mean(v * v)), 'mode': mode}
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        for kind, fn in want.items():
            got = raster.block_overview(a, f, kind)
            ref = loop(fn)
            assert np.allclose(got, ref, rtol=1e-14, atol=0, equal_nan=True), kind
    near = raster.block_overview(a, f, 'nearest')
    assert near.shape == (N, M) and np.array_equal(near[:4, :5], a[1:11:3, 1:14:3], equal_nan=True)
    assert np.array_equal(raster.block_overview(a, f, 'average'), raster.block_mean_overview(a, f, like_reference=False), equal_nan=True)
    with pytest.raises(NotImplementedError):
        raster.block_overview(a, f, 'lanczos')


def test_writer_nits_metadata_escaping_sum_overview_classic_limit(tmp_path):
    """(1) GDAL_METADATA items are XML-escaped; (2) a 'sum' overview block that holds no data stays NaN like the other
    block statistics; (3) a classic TIFF that would pass 4 GiB is refused with the bigtiff hint, not with struct.error."""
    a = np.arange(36, dtype=np.float64).reshape(6, 6)
    fn = str(tmp_path / 'm.tif')
    raster.write_geotiff(fn, a, (1.0, 0.0, 0.0, 0.0, -1.0, 6.0), tags={'note': 'a<b & "c"', 'k&y': '1'})
    raw = open(fn, 'rb').read()
    assert b'<Item name="note">a&lt;b &amp; "c"</Item>' in raw and b'<Item name="k&amp;y">1</Item>' in raw
    assert np.array_equal(raster.read_geotiff(fn).array, a)
    b = a.copy(); b[:3, :3] = np.nan
    s = raster.block_overview(b, 3, 'sum')
    assert np.isnan(s[0, 0]) and s[0, 1] == a[:3, 3:].sum() and s[1, 0] == a[3:, :3].sum()
    b[0, 0] = 2.0
    assert raster.block_overview(b, 3, 'sum')[0, 0] == 2.0

    class Big(object):              # a file object that pretends to be just below 4 GiB
        def __init__(self, fh): self.fh = fh
        def tell(self): return self.fh.tell() + (1 << 32) - 64
        def __getattr__(self, k): return getattr(self.fh, k)
        def __enter__(self): return self
        def __exit__(self, *a): self.fh.close()
    import builtins
    real_open = builtins.open
    try:
        builtins.open = lambda f, mode='r', *a, **k: Big(real_open(f, mode, *a, **k)) if 'w' in mode and str(f).endswith('big.tif') else real_open(f, mode, *a, **k)
        with pytest.raises(ValueError, match='bigtiff=True'):
            raster.write_geotiff(str(tmp_path / 'big.tif'), a, (1.0, 0.0, 0.0, 0.0, -1.0, 6.0), bigtiff=False)
    finally:
        builtins.open = real_open


@pytest.mark.parametrize('dtype', ['uint8', 'int16', 'int32', 'float32', 'float64'])
def test_lzw_encoder_emits_libtiffs_stream(dtype, tmp_path):
    """The native TIFF-LZW encoder (csrc/tiff_lzw.cpp; the reference exports with compress='lzw', pydem/process_manager.py:905)
    against libtiff itself: the strip written for an array is byte for byte what Pillow's libtiff writes for the same
    bytes (8-bit view: LZW works on the byte stream), our reader decodes both, and libtiff decodes ours."""
    PIL = pytest.importorskip('PIL')
    from PIL import Image, features
    if not features.check('libtiff'):
        pytest.skip("Pillow without libtiff")
    rng = np.random.default_rng(5)
    n, m = 96, 160
    if dtype.startswith('float'):
        a = np.round(rng.random((n, m)) * 50).astype(dtype) + np.arange(m, dtype=dtype) / 8        # smooth + repeats: long strings
    else:
        a = (rng.integers(0, 7, (n, m)) + np.arange(m) // 3).astype(dtype)
    raw = a.tobytes()
    # ---- libtiff's stream for the same bytes: one strip of an 8-bit image that is the byte view of the array
    view = np.frombuffer(raw, np.uint8).reshape(n, m * a.dtype.itemsize)
    ref_fn = str(tmp_path / 'ref.tif')
    Image.fromarray(view).save(ref_fn, format='TIFF', compression='tiff_lzw', tiffinfo={278: n})       # RowsPerStrip = all rows
    with Image.open(ref_fn) as im:
        offs, cnts = im.tag_v2[273], im.tag_v2[279]
        assert len(offs) == 1
    ref_strip = open(ref_fn, 'rb').read()[offs[0]:offs[0] + cnts[0]]
    mine = raster._lzw_encode(raw)
    assert mine == ref_strip, "LZW stream differs from libtiff's (%d vs %d bytes)" % (len(mine), len(ref_strip))
    assert raster._lzw_decode(mine, len(raw)) == raw
    # ---- a whole file: ours decoded by libtiff and by our reader
    fn = str(tmp_path / 'mine.tif')
    raster.write_geotiff(fn, a, (1.0, 0.0, 0.0, 0.0, -1.0, float(n)), compress='lzw')
    assert np.array_equal(raster.read_geotiff(fn).array, a)
    if dtype in ('uint8', 'int16', 'int32', 'float32'):
        with Image.open(fn) as im:
            assert np.array_equal(np.asarray(im), a)
    # tiled, with the table filling up several times (incompressible noise) and an empty payload
    noise = rng.integers(0, 256, 70000, dtype=np.uint8).tobytes()
    assert raster._lzw_decode(raster._lzw_encode(noise), len(noise)) == noise
    assert raster._lzw_decode(raster._lzw_encode(b''), 0) == b''
    fn2 = str(tmp_path / 'tiled.tif')
    raster.write_geotiff(fn2, a, (1.0, 0.0, 0.0, 0.0, -1.0, float(n)), compress='lzw', tile=32, bigtiff=True)
    assert np.array_equal(raster.read_geotiff(fn2).array, a)


@pytest.mark.parametrize('kind', ['zeros_then_binary', 'zeros_then_4sym', 'periodic_then_binary', 'constant_then_noise'])
def test_lzw_encoder_follows_libtiffs_ratio_checkpoints(kind, tmp_path):
    """libtiff's encoder watches its compression ratio every 10000 input bytes and emits a ClearCode when it stopped
    improving (tif_lzw.c: CHECK_GAP / CALCRATIO) -- e.g. a no-data area followed by low-entropy terrain inside one block (with
    byte noise the table fills up and restarts the count before a checkpoint is reached).  The first three payloads make
    libtiff reset on the ratio (an encoder without the checkpoints emits a different stream for them); the native encoder's
    stream is libtiff's, byte for byte."""
    PIL = pytest.importorskip('PIL')
    from PIL import Image, features
    if not features.check('libtiff'):
        pytest.skip("Pillow without libtiff")
    rng = np.random.default_rng(3)
    m = 512
    if kind == 'zeros_then_binary':
        v = np.concatenate([np.zeros(60 * m, np.uint8), rng.integers(0, 2, 300 * m, dtype=np.uint8)])
    elif kind == 'zeros_then_4sym':
        v = np.concatenate([np.zeros(60 * m, np.uint8), rng.integers(0, 4, 300 * m, dtype=np.uint8)])
    elif kind == 'periodic_then_binary':
        v = np.concatenate([np.tile(np.arange(16, dtype=np.uint8), 60 * 32), rng.integers(0, 2, 200 * m, dtype=np.uint8)])
    else:
        v = np.concatenate([np.zeros(85 * m, np.uint8), rng.integers(0, 256, 171 * m, dtype=np.uint8)])
    v = v.reshape(-1, m)
    n = v.shape[0]
    raw = v.tobytes()
    ref_fn = str(tmp_path / 'ref.tif')
    Image.fromarray(v).save(ref_fn, format='TIFF', compression='tiff_lzw', tiffinfo={278: n})
    with Image.open(ref_fn) as im:
        offs, cnts = im.tag_v2[273], im.tag_v2[279]
        assert len(offs) == 1
    ref_strip = open(ref_fn, 'rb').read()[offs[0]:offs[0] + cnts[0]]
    mine = raster._lzw_encode(raw)
    assert mine == ref_strip, "LZW stream differs from libtiff's (%d vs %d bytes)" % (len(mine), len(ref_strip))
    assert raster._lzw_decode(mine, len(raw)) == raw
