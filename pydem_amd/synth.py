"""Deterministic synthetic DEM tiles (inputs only -- no expected outputs).

The generators are built from 32-bit integer hashing and IEEE-754 double
`+ - *` in a fixed order (one division by a constant at the end), so numpy here,
the C oracle (`oracle/pydem_oracle.c: oracle_synth_fractal`) and the HIP kernel
(`pydem_amd/csrc/synth.hip`) produce bit-identical tiles from the same
(seed, origin, shape).  No FFT and no libm calls: `np.fft` output changes with
the SIMD dispatch of the machine.

`cone` restates the *inputs* of the reference's synthetic cone
(reference: pydem/utils_test_pydem.py:98-103 and :422 `np.mgrid[-1:1:NNj, -1:1:NNj]`);
the expected outputs of that file are not reproduced here.
"""
import numpy as np

# value-noise constants (shared verbatim with oracle/pydem_oracle.c and csrc/synth.hip)
_K1 = np.uint32(0x9E3779B1)
_K2 = np.uint32(0x85EBCA77)
_K3 = np.uint32(0xC2B2AE3D)
_M1 = np.uint32(0x2C1B3C6D)
_M2 = np.uint32(0x297A2D39)
GAIN = 0.57            # amplitude ratio between successive octaves (~ Hurst 0.8)
TOP_SHIFT = 12         # coarsest lattice cell = 2**12 pixels
N_OCTAVES = 12         # finest lattice cell = 2 pixels


def _hash01(ix, iy, seed):
    """uint32 lattice hash -> double in [0, 1) (exact: h * 2**-32)."""
    with np.errstate(over='ignore'):
        h = (ix * _K1) ^ (iy * _K2) ^ (seed * _K3)
        h = h ^ (h >> np.uint32(15))
        h = h * _M1
        h = h ^ (h >> np.uint32(12))
        h = h * _M2
        h = h ^ (h >> np.uint32(15))
    return h.astype(np.float64) * 2.0 ** -32


def fractal_unit(n_rows, n_cols, seed=0, row0=0, col0=0,
                 n_octaves=N_OCTAVES, top_shift=TOP_SHIFT):
    """Multi-octave value noise in [0, 1), evaluated at global pixel coordinates
    (row0 + i, col0 + j) so neighbouring tiles of a mosaic are continuous."""
    gi = (np.arange(n_rows, dtype=np.int64) + row0).astype(np.uint32)[:, None]
    gj = (np.arange(n_cols, dtype=np.int64) + col0).astype(np.uint32)[None, :]
    z = np.zeros((n_rows, n_cols), np.float64)
    amp = 1.0
    norm = 0.0
    for o in range(n_octaves):
        s = np.uint32(top_shift - o)
        mask = np.uint32((1 << int(s)) - 1)
        inv = 2.0 ** -int(s)
        iy = gi >> s
        ix = gj >> s
        fy = (gi & mask).astype(np.float64) * inv
        fx = (gj & mask).astype(np.float64) * inv
        ty = (fy * fy) * (3.0 - 2.0 * fy)
        tx = (fx * fx) * (3.0 - 2.0 * fx)
        with np.errstate(over='ignore'):
            sd = np.uint32((seed * 1000003 + o) & 0xFFFFFFFF)
        one = np.uint32(1)
        v00 = _hash01(ix, iy, sd)
        v10 = _hash01(ix + one, iy, sd)
        v01 = _hash01(ix, iy + one, sd)
        v11 = _hash01(ix + one, iy + one, sd)
        a = v00 + tx * (v10 - v00)
        b = v01 + tx * (v11 - v01)
        n = a + ty * (b - a)
        z = z + amp * n
        norm = norm + amp
        amp = amp * GAIN
    return z / norm


def fractal(n_rows, n_cols, seed=0, row0=0, col0=0, zmin=1.0, zrange=1000.0, **kw):
    """fp64 fractal tile in [zmin, zmin + zrange) metres."""
    return zmin + zrange * fractal_unit(n_rows, n_cols, seed, row0, col0, **kw)


def srtm_int16(n_rows, n_cols, seed=3, row0=0, col0=0, zrange=3000.0, lake_level=900,
               **kw):
    """SRTM-like int16 tile: fractal scaled to [0, zrange] m, rounded to integer
    metres, with everything below `lake_level` flooded to that exact level
    (large exact plateaus) -- BASELINE.json config 5 input."""
    z = np.rint(zrange * fractal_unit(n_rows, n_cols, seed, row0, col0, **kw))
    z = np.maximum(z, float(lake_level))
    return z.astype(np.int16)


def cone(nn):
    """The reference's synthetic cone input: x,y on mgrid[-1:1:nn*1j]^2,
    z = 1 - sqrt(x^2+y^2)/sqrt(2)  (utils_test_pydem.py:98-103, :422)."""
    y, x = np.mgrid[-1:1:nn * 1j, -1:1:nn * 1j]
    return 1 - np.sqrt(y ** 2 + x ** 2) / np.sqrt(2.)


def cone_scaled(nn):
    """case_cone_scaled input (utils_test_pydem.py:127-130): cone shifted to be >= 0."""
    z = cone(nn)
    return z - z.ravel().min()


def chunk_edges(nn, n_chunks, overlap):
    """Start/stop indices of `n_chunks` tiles along an axis of length nn with `overlap` shared pixels."""
    size = int(np.ceil(nn / n_chunks))
    lo = np.arange(0, nn - overlap, size)
    lo[1:] -= overlap // 2
    hi = np.arange(0, nn - overlap, size)
    hi[:-1] = hi[1:] + int(np.ceil(overlap / 2))
    hi[-1] = nn
    return lo, np.minimum(hi, nn)


def split_mosaic(raster, ny_grid, nx_grid, overlap, lat=(46.0, 45.0), lon=(-73.0, -72.0)):
    """Cut one raster into overlapping tiles with pixel-centred geographic bounds; yields
    (elev, (left, bottom, right, top)) in row-major tile order."""
    ni, nj = raster.shape
    la = np.linspace(lat[0], lat[1], ni)
    lo = np.linspace(lon[0], lon[1], nj)
    te_, be_ = chunk_edges(ni, ny_grid, overlap)
    le_, re_ = chunk_edges(nj, nx_grid, overlap)
    for te, be in zip(te_, be_):
        for le, re in zip(le_, re_):
            h = abs(la[te] - la[be - 1]) / (be - te - 1.0)
            w = abs(lo[le] - lo[re - 1]) / (re - le - 1.0)
            top, left = la[te] + h / 2, lo[le] - w / 2
            yield raster[te:be, le:re].copy(), (left, top - h * (be - te), left + w * (re - le), top)
