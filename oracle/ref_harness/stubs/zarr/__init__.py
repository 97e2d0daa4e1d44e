"""In-memory stand-in for `zarr` (not installed).  TEST INFRASTRUCTURE ONLY.  Arrays are numpy
arrays registered by path for the life of the process (the reference is run with n_workers=1)."""
import os

import numpy as np

_STORE = {}


def reset():
    _STORE.clear()


class Array(object):
    """Array with zarr's value semantics: indexing returns a COPY, assignment writes through."""

    def __init__(self, shape, fill_value, dtype):
        self._a = np.full(tuple(shape), fill_value if fill_value is not None else 0, dtype=dtype or 'float64')
        self.shape = self._a.shape
        self.dtype = self._a.dtype

    def __getitem__(self, key):
        return np.array(self._a[key])

    def __setitem__(self, key, value):
        self._a[key] = value

    def __array__(self, dtype=None, copy=None):
        return np.array(self._a, dtype=dtype)


class Group(object):
    def __init__(self, path):
        self.path = path

    def __getitem__(self, name):
        return open(os.path.join(self.path, name))

    def __contains__(self, name):
        return os.path.join(self.path, name) in _STORE


def open(path, mode='a', shape=None, chunks=None, dtype=None, fill_value=0, **kw):
    path = os.path.normpath(path)
    if path in _STORE and not os.path.isdir(path):
        del _STORE[path]          # the directory was removed (tests rmtree between cases): stale entry
    if path in _STORE:
        return _STORE[path]
    if shape is None:
        return Group(path)
    arr = Array(shape, fill_value, dtype)
    _STORE[path] = arr
    os.makedirs(path, exist_ok=True)      # marker on disk so a later rmtree invalidates the entry
    return arr
