"""Whole-plane transfers and plane reuse (csrc/tile.hip: tile_plane_copy, plane_take / plane_give).

Planes of 32 MiB and more travel through pinned chunks on several host threads; the planes of a destroyed tile go to the
next tile of the same shape with whatever they held.  Bar: bytes in = bytes out for every dtype the upload converts, at
sizes that are not multiples of the chunk, and a second tile on reused planes computes what a first one on fresh planes
did (no stage may rely on what a plane held before)."""
import gc

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('shape,dtype', [((2100, 2050), np.float64),      # 34.4 MB: four chunks and a ragged fifth
                                         ((4200, 4100), np.int16),        # 34.4 MB of int16: chunked upload + conversion on the device
                                         ((3000, 3000), np.float32),
                                         ((300, 200), np.float64)])       # below the threshold: the plain copy
def test_plane_round_trip(shape, dtype):
    from pydem_amd import _ffi
    rng = np.random.default_rng(5)
    if np.dtype(dtype).kind == 'f':
        z = rng.standard_normal(shape).astype(dtype)
        z[7, 11] = np.nan
    else:
        z = rng.integers(-3000, 9000, size=shape).astype(dtype)
    t = _ffi.Tile(shape[0], shape[1])
    t.upload(_ffi.ELEV, z)
    back = t.download(_ffi.ELEV)
    assert back.dtype == np.float64 and back.shape == shape
    assert np.array_equal(back, z.astype(np.float64), equal_nan=True)
    _ffi.release_scratch()                              # drops the pinned chunks and streams: the next transfer sets them up again
    assert np.array_equal(t.download(_ffi.ELEV), z.astype(np.float64), equal_nan=True)
    # a uint8 plane (1 byte per cell) of the same tile
    m = (rng.random(shape) < 0.3).astype(np.uint8)
    t.upload(_ffi.FLATS, m)
    assert np.array_equal(t.download(_ffi.FLATS), m)


def test_second_tile_on_reused_planes_matches_the_first():
    from pydem_amd import DEMProcessor, synth, _ffi
    n = 1536                                            # planes of 18.9 MB (fp64) / 2.4 MB (bytes): all above the 1-MiB floor of the free lists
    za, zb = synth.fractal(n, n, seed=11), synth.fractal(n, n, seed=12)

    def run(z):
        dp = DEMProcessor(elev=z, dX=30.0, dY=30.0, fill_flats=True, drain_pits_path=True, drain_pits=True)
        twi = dp.calc_twi()
        out = {k: np.array(getattr(dp, k)) for k in ('mag', 'direction', 'uca', 'edge_todo', 'edge_done', 'flats')}
        out['twi'] = np.array(twi)
        del dp
        gc.collect()
        return out

    first_b = run(zb)                                   # fresh planes (or whatever earlier tests left: released below)
    _ffi.load().pydem_hip_release_scratch()             # empty free lists: the next tile maps new memory
    fresh_b = run(zb)
    run(za)                                             # leaves its planes, full of another tile's values, on the free lists
    reused_b = run(zb)
    for k in fresh_b:
        assert np.array_equal(fresh_b[k], reused_b[k], equal_nan=True), k
        assert np.array_equal(fresh_b[k], first_b[k], equal_nan=True), k
