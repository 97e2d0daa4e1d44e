"""In-memory stand-in for `rasterio` (not installed).  TEST INFRASTRUCTURE ONLY -- lets the
unmodified reference ProcessManager run in the build container so directory-flow goldens can be
captured.  "Files" are pickles holding the array, an affine transform and a CRS flag; every CRS
reports is_projected=True so the reference takes pixel sizes as spacing (utils.py:132-137) --
the reference's own tests then overwrite the spacing with 1 (process_manager.DEBUG)."""
import builtins
import pickle
from collections import namedtuple

import numpy as np

BoundingBox = namedtuple('BoundingBox', 'left bottom right top')


class Affine(object):
    def __init__(self, a, b, c, d, e, f):
        self.a, self.b, self.c, self.d, self.e, self.f = a, b, c, d, e, f

    @classmethod
    def translation(cls, x, y):
        return cls(1.0, 0.0, x, 0.0, 1.0, y)

    @classmethod
    def scale(cls, sx, sy):
        return cls(sx, 0.0, 0.0, 0.0, sy, 0.0)

    @classmethod
    def from_gdal(cls, c, a, b, f, d, e):
        return cls(a, b, c, d, e, f)

    def __mul__(self, o):
        return Affine(self.a * o.a + self.b * o.d, self.a * o.b + self.b * o.e, self.a * o.c + self.b * o.f + self.c,
                      self.d * o.a + self.e * o.d, self.d * o.b + self.e * o.e, self.d * o.c + self.e * o.f + self.f)


class _T(object):
    Affine = Affine


transform = _T()


class _CRS(object):
    is_projected = True

    def __init__(self, name):
        self.name = name

    def to_wkt(self):
        return 'PROJCS["fake",SPHEROID["WGS 84"]]'


class _Dataset(object):
    def __init__(self, fn, mode='r', **kw):
        self.fn, self.mode, self.kw = fn, mode, kw
        if mode == 'r':
            with builtins.open(fn, 'rb') as f:
                d = pickle.load(f)
            self._data, self.transform, self.crs = d['data'], Affine(*d['transform']), _CRS(d['crs'])
            self.shape = self._data.shape
            t = self.transform
            h, w = self.shape
            self.bounds = BoundingBox(t.c, t.f + t.e * h, t.c + t.a * w, t.f)

    def read(self, band=1):
        return self._data.copy()

    def write(self, data, band=1):
        t = self.kw['transform']
        with builtins.open(self.fn, 'wb') as f:
            pickle.dump(dict(data=np.array(data), transform=(t.a, t.b, t.c, t.d, t.e, t.f), crs=str(self.kw.get('crs'))), f)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def open(fn, mode='r', **kw):
    mode = kw.pop('mode', mode)
    return _Dataset(fn, mode, **kw)
