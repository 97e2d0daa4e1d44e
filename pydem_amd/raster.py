"""Minimal GeoTIFF reader / writer and WGS-84 pixel spacing -- what the reference gets from rasterio and geopy.

The reference opens every tile with rasterio (pydem/utils.py:43-51 `read_raster`, `dem_processor_from_raster_kwargs`)
and derives the per-row spacing `dX, dY, dX2, dY2` with `geopy.distance` on the tile's geotransform
(`mk_dx_dy_from_geotif_layer`, pydem/utils.py:127-174).  Neither package is part of the accelerated path, and
neither is needed for it: this module reads the single-band GeoTIFFs DEM tiles come in (strips or tiles;
uncompressed, LZW, Deflate or PackBits; horizontal predictor; 8/16/32-bit integers, float32/float64; little or
big endian; classic and BigTIFF), extracts geotransform / model type / ellipsoid / nodata from the GeoTIFF tags, and
reproduces the reference's spacing rules with Vincenty's inverse formula on the ellipsoid (agrees with
geographiclib -- what geopy calls -- to ~1e-10 relative on pixel-sized lines; the results of the terrain path are
compared at 1e-6).  If rasterio is importable it is NOT used: one code path, testable here.

`write_geotiff` is the counterpart (strips or tiles, uncompressed, LZW or Deflate, classic or BigTIFF, overviews) used by the
tests and by `ProcessManager.save_non_overlap_data_geotiff` / `save_geotiff`.
"""
import struct
import zlib

import numpy as np

# ---------------------------------------------------------------------------------------------------------------
# TIFF decoding
# ---------------------------------------------------------------------------------------------------------------
_TYPE_FMT = {1: 'B', 2: 'c', 3: 'H', 4: 'I', 5: 'II', 6: 'b', 7: 'B', 8: 'h', 9: 'i', 10: 'ii', 11: 'f', 12: 'd', 16: 'Q', 17: 'q',
             18: 'Q'}
_ELLIPSOIDS = {            # name -> (semi-major axis [m], flattening); keys as geopy spells them (utils.py:140-151)
    'WGS-84': (6378137.0, 1 / 298.257223563),
    'GRS-80': (6378137.0, 1 / 298.257222101),
    'Airy (1830)': (6377563.396, 1 / 299.3249646),
    'Intl 1924': (6378388.0, 1 / 297.0),
    'Clarke (1880)': (6378249.145, 1 / 293.465),
    'GRS-67': (6378160.0, 1 / 298.25),
    'Clarke (1866)': (6378206.4, 1 / 294.9786982),
}
_GEOG_CODES = {4326: 'WGS-84', 4269: 'GRS-80', 4258: 'GRS-80', 4267: 'Clarke (1866)', 4230: 'Intl 1924'}
_ELLIPSOID_CODES = {7030: 'WGS-84', 7019: 'GRS-80', 7022: 'Intl 1924', 7001: 'Airy (1830)', 7036: 'GRS-67'}


def _lzw_decode(data, expected):
    """TIFF flavour of LZW (MSB-first codes, 9..12 bits, ClearCode 256, EOI 257, 'early change')."""
    out = bytearray()
    table = None
    nbits = 9
    acc = 0
    have = 0
    prev = None
    pos = 0
    n = len(data)
    while True:
        while have < nbits and pos < n:
            acc = (acc << 8) | data[pos]; pos += 1; have += 8
        if have < nbits:
            break
        code = (acc >> (have - nbits)) & ((1 << nbits) - 1)
        have -= nbits
        if code == 256:
            table = [bytes([i]) for i in range(256)] + [b'', b'']
            nbits = 9
            prev = None
            continue
        if code == 257:
            break
        if table is None:
            raise ValueError("LZW stream does not start with a clear code")
        if prev is None:
            entry = table[code]
        elif code < len(table):
            entry = table[code]
            table.append(prev + entry[:1])
        else:
            entry = prev + prev[:1]
            table.append(entry)
        out += entry
        prev = entry
        if len(table) >= (1 << nbits) - 1 and nbits < 12:
            nbits += 1
        if len(out) >= expected:
            break
    return bytes(out[:expected])


def _packbits_decode(data, expected):
    out = bytearray()
    i = 0
    while i < len(data) and len(out) < expected:
        n = data[i]; i += 1
        if n < 128:
            out += data[i:i + n + 1]; i += n + 1
        elif n > 128:
            out += data[i:i + 1] * (257 - n); i += 1
    return bytes(out[:expected])


class GeoTiff(object):
    """What the terrain path needs from a rasterio dataset: `read()`, `shape`, `transform` (a, b, c, d, e, f),
    `bounds` (left, bottom, right, top), `is_projected`, `ellipsoid`, `nodata`."""

    def __init__(self, array, transform, is_projected, ellipsoid, nodata):
        self.array = array
        self.shape = array.shape
        self.transform = transform
        self.is_projected = is_projected
        self.ellipsoid = ellipsoid
        self.nodata = nodata
        a, b, c, d, e, f = transform
        h, w = array.shape
        xs = [c, c + a * w]
        ys = [f, f + e * h]
        self.bounds = (min(xs), min(ys), max(xs), max(ys))

    def read(self, band=1):
        return self.array


def read_geotiff(path, page=0):
    """`page` = index in the file's IFD chain: 0 is the full-resolution raster, the overviews written by write_geotiff follow."""
    with open(path, 'rb') as fh:
        buf = fh.read()
    if buf[:2] == b'II':
        bo = '<'
    elif buf[:2] == b'MM':
        bo = '>'
    else:
        raise ValueError("%s is not a TIFF file" % path)
    magic = struct.unpack(bo + 'H', buf[2:4])[0]
    if magic == 42:
        big = False
        ifd = struct.unpack(bo + 'I', buf[4:8])[0]
    elif magic == 43:
        big = True
        ifd = struct.unpack(bo + 'Q', buf[8:16])[0]
    else:
        raise ValueError("%s: unknown TIFF version %d" % (path, magic))
    for _ in range(int(page)):                                          # walk the chain of image file directories
        if big:
            cnt_ = struct.unpack(bo + 'Q', buf[ifd:ifd + 8])[0]
            ifd = struct.unpack(bo + 'Q', buf[ifd + 8 + 20 * cnt_: ifd + 16 + 20 * cnt_])[0]
        else:
            cnt_ = struct.unpack(bo + 'H', buf[ifd:ifd + 2])[0]
            ifd = struct.unpack(bo + 'I', buf[ifd + 2 + 12 * cnt_: ifd + 6 + 12 * cnt_])[0]
        if ifd == 0:
            raise IndexError("%s has no page %d" % (path, page))
    if big:
        n = struct.unpack(bo + 'Q', buf[ifd:ifd + 8])[0]
        base, esz, cnt_fmt, inline = ifd + 8, 20, 'Q', 8
    else:
        n = struct.unpack(bo + 'H', buf[ifd:ifd + 2])[0]
        base, esz, cnt_fmt, inline = ifd + 2, 12, 'I', 4
    tags = {}
    for k in range(n):
        e = buf[base + k * esz: base + (k + 1) * esz]
        tag, typ = struct.unpack(bo + 'HH', e[:4])
        cnt = struct.unpack(bo + cnt_fmt, e[4:4 + inline])[0]
        fmt = _TYPE_FMT.get(typ)
        if fmt is None:
            continue
        size = struct.calcsize(bo + fmt) * cnt
        if size <= inline:
            raw = e[4 + inline: 4 + inline + size]
        else:
            off = struct.unpack(bo + cnt_fmt, e[4 + inline: 4 + 2 * inline])[0]
            raw = buf[off:off + size]
        if typ == 2:
            tags[tag] = raw.split(b'\x00')[0].decode('latin-1')
        else:
            vals = struct.unpack(bo + fmt * cnt if len(fmt) == 1 else bo + fmt * cnt, raw)
            if typ in (5, 10):
                vals = tuple(vals[i] / vals[i + 1] if vals[i + 1] else 0.0 for i in range(0, len(vals), 2))
            tags[tag] = vals
    w, h = tags[256][0], tags[257][0]
    spp = tags.get(277, (1,))[0]
    if spp != 1:
        raise NotImplementedError("%s: %d samples per pixel (single-band DEMs only)" % (path, spp))
    bits = tags.get(258, (1,))[0]
    fmt_code = tags.get(339, (1,))[0]
    comp = tags.get(259, (1,))[0]
    pred = tags.get(317, (1,))[0]
    kind = {1: 'u', 2: 'i', 3: 'f'}.get(fmt_code)
    if kind is None or bits not in (8, 16, 32, 64) or (kind == 'f' and bits < 32):
        raise NotImplementedError("%s: sample format %d with %d bits" % (path, fmt_code, bits))
    dt = np.dtype(bo + kind + str(bits // 8))

    def inflate(chunk, nbytes):
        if comp == 1:
            return chunk[:nbytes]
        if comp == 5:
            return _lzw_decode(chunk, nbytes)
        if comp in (8, 32946):
            return zlib.decompress(chunk)[:nbytes]
        if comp == 32773:
            return _packbits_decode(chunk, nbytes)
        raise NotImplementedError("%s: TIFF compression %d" % (path, comp))

    def unpredict(block):
        if pred == 1:
            return block
        if pred == 2 and kind in 'ui':
            return np.cumsum(block, axis=1, dtype=block.dtype)          # wraps like the encoder's differences
        raise NotImplementedError("%s: predictor %d for %s samples" % (path, pred, dt))

    out = np.empty((h, w), dt.newbyteorder('='))
    if 322 in tags:                                                     # tiled
        tw, th = tags[322][0], tags[323][0]
        offs, cnts = tags[324], tags[325]
        per_row = (w + tw - 1) // tw
        for t, (o, c) in enumerate(zip(offs, cnts)):
            r0, c0 = (t // per_row) * th, (t % per_row) * tw
            blk = np.frombuffer(inflate(buf[o:o + c], tw * th * dt.itemsize), dt).reshape(th, tw)
            blk = unpredict(blk)
            out[r0:r0 + th, c0:c0 + tw] = blk[:min(th, h - r0), :min(tw, w - c0)]
    else:
        rps = tags.get(278, (h,))[0]
        offs, cnts = tags[273], tags.get(279)
        for s, o in enumerate(offs):
            r0 = s * rps
            rows = min(rps, h - r0)
            c = cnts[s] if cnts else rows * w * dt.itemsize
            blk = np.frombuffer(inflate(buf[o:o + c], rows * w * dt.itemsize), dt).reshape(rows, w)
            out[r0:r0 + rows] = unpredict(blk)
    # ---- georeferencing
    if 34264 in tags:                                                   # ModelTransformationTag (4x4)
        m = tags[34264]
        transform = (m[0], m[1], m[3], m[4], m[5], m[7])
    elif 33550 in tags and 33922 in tags:                               # pixel scale + tie point
        sx, sy = tags[33550][0], tags[33550][1]
        i, j, _, x, y, _ = tags[33922][:6]
        transform = (sx, 0.0, x - i * sx, 0.0, -sy, y + j * sy)
    else:
        transform = (1.0, 0.0, 0.0, 0.0, -1.0, float(h))
    projected, ellipsoid = False, 'WGS-84'
    if 34735 in tags:
        keys = tags[34735]
        for k in range(1, keys[3] + 1):
            key, loc, cnt, val = keys[4 * k: 4 * k + 4]
            if key == 1024 and loc == 0:
                projected = (val == 1)
            elif key == 2048 and loc == 0:
                ellipsoid = _GEOG_CODES.get(val, ellipsoid)
            elif key == 2056 and loc == 0:
                ellipsoid = _ELLIPSOID_CODES.get(val, ellipsoid)
    else:
        projected = True                                                # no geo keys: plain pixel coordinates
    nodata = None
    if 42113 in tags:
        try:
            nodata = float(tags[42113])
        except ValueError:
            nodata = None
    # pixel-is-point rasters (GTRasterTypeGeoKey 1025 == 2) anchor the tie point at the pixel centre
    if 34735 in tags:
        keys = tags[34735]
        for k in range(1, keys[3] + 1):
            key, loc, cnt, val = keys[4 * k: 4 * k + 4]
            if key == 1025 and loc == 0 and val == 2:
                a, b, c, d, e, f = transform
                transform = (a, b, c - a / 2, d, e, f - e / 2)
    return GeoTiff(out, transform, projected, ellipsoid, nodata)


def _lzw_encode(payload):
    """TIFF LZW through the native encoder (csrc/tiff_lzw.cpp: libtiff's stream for the same bytes)."""
    import ctypes as C
    from . import _ffi
    lib = _ffi.load()
    n = len(payload)
    cap = n + n // 2 + 64
    dst = (C.c_uint8 * cap)()
    out_n = C.c_int64(0)
    src = (C.c_uint8 * max(n, 1)).from_buffer_copy(payload if n else b'\0')
    _ffi.check(lib.pydem_tiff_lzw_encode(src, n, dst, cap, C.byref(out_n)))
    return bytes(memoryview(dst)[:out_n.value])


def write_geotiff(path, array, transform, projected=False, nodata=None, compress=False, overviews=(), tile=None, bigtiff=False,
                  tags=None):
    """Single-band little-endian GeoTIFF.  `transform` = (a, b, c, d, e, f).

    Layout: one strip per image (`tile=None`) or square tiles of `tile` pixels (a multiple of 16; the reference writes
    512 x 512 blocks, pydem/process_manager.py:906-913); `compress`: False / None = none, 'lzw' = TIFF LZW (what the
    reference asks rasterio for, :905: libtiff's stream through the native encoder csrc/tiff_lzw.cpp), True / 'deflate' =
    zlib; classic TIFF (offsets of 32 bits: < 4 GiB, checked) or BigTIFF (`bigtiff=True`, 64-bit offsets, what the
    reference always writes).
    `overviews`: reduced-resolution copies (arrays, largest first) written as further image file directories behind the
    full-resolution one (NewSubfileType = 1: what GDAL / rasterio list as the dataset's overviews).
    `tags`: dict of GDAL metadata items (name -> value) stored in the GDAL_METADATA tag of the first image (the reference's
    `update_tags`, :925, :931).
    The pixel data are streamed to the file image by image, block by block; the directories follow at the end."""
    a, b, c, d, e, f = transform
    if compress in (None, False, 0, 'none'):
        codec = 1
    elif compress == 'lzw':
        codec = 5
    elif compress in (True, 'deflate'):
        codec = 8
    else:
        raise ValueError("compress must be None, 'lzw' or 'deflate', not %r" % (compress,))
    base_dtype = np.asarray(array).dtype
    images = [np.ascontiguousarray(array)]
    for o in overviews:
        o = np.asarray(o)
        if base_dtype.kind in 'iu' and o.dtype.kind == 'f':
            # block means of an integer raster: a block without valid cells is NaN, which no integer can carry -- it becomes the
            # nodata value (tag 42113 announces it) or the export is refused; the other means are truncated like the base cast.
            # (Chained levels are means of means and edge blocks are partial: not GDAL's 'average' on the last row / column.)
            bad = np.isnan(o)
            if bad.any():
                if nodata is None:
                    raise ValueError("overview blocks without data in an integer GeoTIFF need a nodata value")
                o = np.where(bad, nodata, o)
        images.append(np.ascontiguousarray(o).astype(base_dtype))
    if tile is not None and (int(tile) <= 0 or int(tile) % 16):
        raise ValueError("TIFF tile edges are multiples of 16, not %r" % (tile,))
    h0, w0 = images[0].shape
    off_t, off_fmt = (16, 'Q') if bigtiff else (4, 'I')
    with open(path, 'wb') as fh:
        fh.write(b'II' + (struct.pack('<HHHQ', 43, 8, 0, 0) if bigtiff else struct.pack('<HI', 42, 0)))     # first IFD offset: patched below
        dirs = []          # per image: sorted entries (tag, type, count, packed bytes)
        for level, arr in enumerate(images):
            if arr.dtype.byteorder == '>':
                arr = arr.astype(arr.dtype.newbyteorder('<'))
            kind = {'u': 1, 'i': 2, 'f': 3}[arr.dtype.kind]
            h, w = arr.shape
            offs, cnts = [], []

            def put(block):
                payload = block.tobytes()
                if codec == 5:
                    payload = _lzw_encode(payload)
                elif codec == 8:
                    payload = zlib.compress(payload, 6)
                if fh.tell() % 2:
                    fh.write(b'\x00')
                if not bigtiff and fh.tell() + len(payload) >= 1 << 32:     # (before the offsets are packed as 32-bit words)
                    raise ValueError("write_geotiff: the file would exceed the 4 GiB of classic TIFF (use bigtiff=True)")
                offs.append(fh.tell()); cnts.append(len(payload))
                fh.write(payload)
            if tile is None:
                put(arr)
            else:
                t = int(tile)
                for r0 in range(0, h, t):
                    for c0 in range(0, w, t):
                        blk = np.zeros((t, t), arr.dtype)                     # edge tiles are padded to the full size
                        part = arr[r0:r0 + t, c0:c0 + t]
                        blk[:part.shape[0], :part.shape[1]] = part
                        put(blk)
            entries = []

            def add(tag, typ, values, entries=entries):
                fmt = _TYPE_FMT[typ]
                if typ == 2:
                    raw = values.encode('latin-1') + b'\x00'
                    entries.append((tag, typ, len(raw), raw))
                else:
                    entries.append((tag, typ, len(values), struct.pack('<' + fmt * len(values), *values)))

            if level > 0:
                add(254, 4, [1])                                            # reduced-resolution version of another image
            add(256, 4, [w]); add(257, 4, [h]); add(258, 3, [arr.dtype.itemsize * 8]); add(259, 3, [codec])
            add(262, 3, [1]); add(277, 3, [1]); add(339, 3, [kind])
            if tile is None:
                add(273, off_t, offs); add(278, 4, [h]); add(279, off_t, cnts)
            else:
                add(322, 4, [int(tile)]); add(323, 4, [int(tile)]); add(324, off_t, offs); add(325, off_t, cnts)
            # (an overview covers the same ground with fewer, larger pixels)
            add(33550, 12, [a * w0 / w, -e * h0 / h, 0.0]); add(33922, 12, [0.0, 0.0, 0.0, c, f, 0.0])
            add(34735, 3, [1, 1, 0, 3, 1024, 0, 1, 1 if projected else 2, 1025, 0, 1, 1, 2048, 0, 1, 4326])
            if tags and level == 0:
                from xml.sax.saxutils import escape, quoteattr
                items = ''.join('<Item name=%s>%s</Item>' % (quoteattr(str(k)), escape(str(v))) for k, v in tags.items())
                add(42112, 2, '<GDALMetadata>' + items + '</GDALMetadata>')
            if nodata is not None:
                add(42113, 2, repr(float(nodata)))
            entries.sort(key=lambda t: t[0])
            dirs.append(entries)
        # ---- the image file directories: count | entries | next | long values
        inline = 8 if bigtiff else 4
        first = None
        for level, entries in enumerate(dirs):
            if fh.tell() % 2:
                fh.write(b'\x00')
            ifd_off = fh.tell()
            if first is None:
                first = ifd_off
            ifd_len = (8 + 20 * len(entries) + 8) if bigtiff else (2 + 12 * len(entries) + 4)
            extra_off = ifd_off + ifd_len
            extra, body = b'', b''
            for tag, typ, cnt, raw in entries:
                if len(raw) <= inline:
                    val = raw + b'\x00' * (inline - len(raw))
                else:
                    val = struct.pack('<' + off_fmt, extra_off + len(extra))
                    extra += raw + (b'\x00' if len(raw) % 2 else b'')
                body += struct.pack('<HHQ' if bigtiff else '<HHI', tag, typ, cnt) + val
            end = extra_off + len(extra)
            end += end % 2
            nxt = end if level + 1 < len(dirs) else 0
            if not bigtiff and end >= 1 << 32:
                raise ValueError("write_geotiff: the file would exceed the 4 GiB of classic TIFF (use bigtiff=True)")
            fh.write(struct.pack('<Q' if bigtiff else '<H', len(entries)) + body + struct.pack('<' + off_fmt, nxt) + extra)
        fh.seek(8 if bigtiff else 4)
        fh.write(struct.pack('<' + off_fmt, first))


# ---------------------------------------------------------------------------------------------------------------
# overviews (reduced-resolution copies)
# ---------------------------------------------------------------------------------------------------------------
def block_mean_overview(data, factor, like_reference=True):
    """One level of the overview pyramid: every output cell is the mean of a factor x factor block of `data`; the output
    has ceil(n / factor) rows and columns, the last row / column hold the blocks that stick out of the array.

    like_reference=True reproduces calc_overview of the reference (pydem/process_manager.py:317-352) bit for bit -- same
    summation order (numpy's mean over a contiguous run of factor^2 values) and its treatment of the partial blocks: the
    right-hand blocks are means over the columns that are left, the corner is the mean of what is left of it, and the
    BOTTOM row is written with the reference's reshape (:340-342), which averages the leftover rows over every
    (n_cols // factor)-th column instead of over the block's own columns.  like_reference=False gives the plain partial-
    block means there too (what GDAL's 'average' overview resampling does), used for the GeoTIFF overviews."""
    data = np.asarray(data, np.float64)
    f = int(factor)
    n, m = data.shape
    R, C = n // f, m // f
    out = np.zeros((-(-n // f), -(-m // f)), np.float64)
    full = np.ascontiguousarray(data[:R * f, :C * f].reshape(R, f, C, f).transpose(0, 2, 1, 3)).reshape(R, C, f * f)
    out[:R, :C] = full.mean(axis=-1)
    with np.errstate(invalid='ignore'):
        if m > C * f:
            out[:R, C:] = np.ascontiguousarray(data[:R * f, C * f:]).reshape(R, -1).mean(axis=1)[:, None]
        if n > R * f:
            rest = np.ascontiguousarray(data[R * f:, :C * f])
            if like_reference:
                out[R:, :C] = rest.reshape(-1, C).mean(axis=0)[None, :]
            else:
                out[R:, :C] = rest.reshape(n - R * f, C, f).transpose(1, 0, 2).reshape(C, -1).mean(axis=1)[None, :]
        if n > R * f and m > C * f:
            out[R:, C:] = data[R * f:, C * f:].mean()
    return out


# block statistics GDAL / rasterio offer as overview resampling besides 'average' (rasterio.enums.Resampling); the
# interpolating kernels (bilinear, cubic, cubic_spline, lanczos, gauss) are GDAL's own and are not reproduced
_BLOCK_STATS = {
    'max': lambda v: np.nanmax(v, axis=-1), 'min': lambda v: np.nanmin(v, axis=-1), 'med': lambda v: np.nanmedian(v, axis=-1),
    'q1': lambda v: np.nanquantile(v, 0.25, axis=-1), 'q3': lambda v: np.nanquantile(v, 0.75, axis=-1),
    'sum': lambda v: np.where(np.isnan(v).all(axis=-1), np.nan, np.nansum(v, axis=-1)), 'rms': lambda v: np.sqrt(np.nanmean(v * v, axis=-1)),
}
OVERVIEW_KINDS = ('average', 'nearest', 'mode') + tuple(sorted(_BLOCK_STATS))


def block_overview(data, factor, kind='average'):
    """One overview level of `data` with one of OVERVIEW_KINDS: every output cell summarises a factor x factor block (the
    blocks that stick out of the array are partial), NaN = no data is ignored by the statistics.  'average' is
    block_mean_overview(like_reference=False); 'nearest' takes the block's centre cell like GDAL's nearest-neighbour
    decimation; 'mode' the most frequent value (ties: the smallest)."""
    data = np.asarray(data, np.float64)
    f = int(factor)
    if kind == 'average':
        return block_mean_overview(data, f, like_reference=False)
    n, m = data.shape
    N, M = -(-n // f), -(-m // f)
    if kind == 'nearest':
        ii = np.minimum(np.arange(N) * f + f // 2, n - 1)
        jj = np.minimum(np.arange(M) * f + f // 2, m - 1)
        return data[np.ix_(ii, jj)].copy()
    pad = np.full((N * f, M * f), np.nan)
    pad[:n, :m] = data
    blocks = np.ascontiguousarray(pad.reshape(N, f, M, f).transpose(0, 2, 1, 3)).reshape(N, M, f * f)
    if kind == 'mode':
        srt = np.sort(blocks, axis=-1)                      # NaN sorts last
        out = np.full((N, M), np.nan)
        best = np.zeros((N, M), np.int64)
        run = np.ones((N, M), np.int64)
        for k in range(f * f):
            v = srt[..., k]
            if k > 0:
                run = np.where(v == srt[..., k - 1], run + 1, 1)
            take = (run > best) & ~np.isnan(v)
            out = np.where(take, v, out)
            best = np.where(take, run, best)
        return out
    if kind not in _BLOCK_STATS:
        raise NotImplementedError("overview resampling %r (available: %s)" % (kind, ', '.join(OVERVIEW_KINDS)))
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore', RuntimeWarning)         # all-NaN blocks stay NaN
        return _BLOCK_STATS[kind](blocks)


# ---------------------------------------------------------------------------------------------------------------
# geodesic pixel spacing
# ---------------------------------------------------------------------------------------------------------------
def geodesic_m(lat1, lon1, lat2, lon2, ellipsoid='WGS-84'):
    """Geodesic distance in metres between two points (degrees) on the ellipsoid: Vincenty's inverse formula
    (what geopy.distance returned before it switched to geographiclib; the two agree to ~1e-10 relative on
    pixel-sized lines; the iteration converges in 2-3 steps for nearby points and degenerates gracefully on a
    meridian)."""
    a, f = _ELLIPSOIDS[ellipsoid]
    b = a * (1 - f)
    p1, p2 = np.radians(lat1), np.radians(lat2)
    L = np.radians(lon2 - lon1)
    if lat1 == lat2 and L == 0:
        return 0.0
    U1 = np.arctan((1 - f) * np.tan(p1))
    U2 = np.arctan((1 - f) * np.tan(p2))
    sU1, cU1, sU2, cU2 = np.sin(U1), np.cos(U1), np.sin(U2), np.cos(U2)
    lam = L
    for _ in range(200):
        sl, cl = np.sin(lam), np.cos(lam)
        ss = np.hypot(cU2 * sl, cU1 * sU2 - sU1 * cU2 * cl)
        if ss == 0:
            return 0.0
        cs = sU1 * sU2 + cU1 * cU2 * cl
        sig = np.arctan2(ss, cs)
        sa = cU1 * cU2 * sl / ss
        c2a = 1 - sa * sa
        c2m = cs - 2 * sU1 * sU2 / c2a if c2a != 0 else 0.0
        C = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
        new = L + (1 - C) * f * sa * (sig + C * ss * (c2m + C * cs * (-1 + 2 * c2m * c2m)))
        if abs(new - lam) < 1e-15:
            lam = new
            break
        lam = new
    u2 = c2a * (a * a - b * b) / (b * b)
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    ds = B * ss * (c2m + B / 4 * (cs * (-1 + 2 * c2m * c2m) - B / 6 * c2m * (-3 + 4 * ss * ss) * (-3 + 4 * c2m * c2m)))
    return float(b * A * (sig - ds))


def spacing_from_geotransform(n_rows, transform, projected, ellipsoid='WGS-84'):
    """dX, dY (n_rows - 1 values) and dX2, dY2 (n_rows values) exactly as `mk_dx_dy_from_geotif_layer`
    (pydem/utils.py:127-174) builds them: projected rasters use the pixel size; geographic ones measure, per row, the
    geodesic between two points one pixel apart in longitude (dX) and the meridian arc to the next row (dY).  The
    reference anchors the longitude at `transform.d` (the rotation term, 0 for north-up rasters) -- irrelevant, only
    differences enter -- and the latitude at `transform.f`; both quirks are kept."""
    a, b, c, d, e, f = transform
    if projected:
        return (np.ones(n_rows - 1) * a, np.abs(np.ones(n_rows - 1) * e), np.ones(n_rows) * a, np.abs(np.ones(n_rows) * e))
    dx, dy = a, e

    def clip(v):
        return min(max(v, -90.0), 90.0)

    lon, lat = d + dx / 2, f + dy / 2
    dX = np.array([geodesic_m(clip(lat + dy * (j + 1)), lon + dx, clip(lat + dy * (j + 1)), lon, ellipsoid) for j in range(n_rows - 1)])
    dY = np.array([geodesic_m(clip(lat + dy * i), lon, clip(lat + dy * (i + 1)), lon, ellipsoid) for i in range(n_rows - 1)])
    lon, lat = d + dx, f + dy
    dX2 = np.array([geodesic_m(clip(lat + dy * (j + 1)), lon + dx, clip(lat + dy * (j + 1)), lon, ellipsoid) for j in range(n_rows)])
    dY2 = np.array([geodesic_m(clip(lat + dy * i), lon, clip(lat + dy * (i + 1)), lon, ellipsoid) for i in range(n_rows)])
    return dX, dY, dX2, dY2


def dem_processor_from_raster_kwargs(path):
    """The reference's helper of the same name (pydem/utils.py:46-51): constructor arguments of a DEMProcessor for one
    raster tile.  Nodata cells become a masked array like rasterio's `read(masked=...)` users expect downstream."""
    ds = read_geotiff(path)
    dX, dY, dX2, dY2 = spacing_from_geotransform(ds.shape[0], ds.transform, ds.is_projected, ds.ellipsoid)
    return dict(dX=dX, dY=dY, elev=ds.read(1), bounds=ds.bounds, transform=ds.transform, dX2=dX2, dY2=dY2,
                is_projected=ds.is_projected)
