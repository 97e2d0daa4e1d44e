"""Drop-in for the reference's Cython module `pydem.cyfuncs.cyutils` (pydem/cyfuncs/cyutils.pyx):
the same two functions with the same arguments, in-place behaviour and return values, executed on the
GPU (csrc/cyutils.hip).  The accelerated DEMProcessor does not go through these (it never builds the
NN x NN matrix); they keep code written against the Cython boundary working."""
import ctypes as C

import numpy as np

from .. import _ffi

_P = C.c_void_p


def _u8(a, name):
    a = np.asarray(a)
    if a.dtype == np.bool_:
        a = a.view(np.uint8)
    if a.dtype != np.uint8 or not a.flags['C_CONTIGUOUS'] or not a.flags['WRITEABLE']:
        raise TypeError("%s must be a writable contiguous bool/uint8 array (it is updated in place)" % name)
    return a


def _i32(a, name, size=None, at_least=None):
    a = np.ascontiguousarray(a, np.int32)
    if a.ndim != 1 or (size is not None and a.size != size) or (at_least is not None and a.size < at_least):
        raise ValueError("%s has %d entries, expected %s" % (name, a.size, size if size is not None else ">= %d" % at_least))
    return a


def _f64_inplace(a, name, size):
    """What the typed Cython signature accepts for an array it updates in place (cyutils.pyx:78-83)."""
    if not (isinstance(a, np.ndarray) and a.dtype == np.float64 and a.flags['C_CONTIGUOUS'] and a.flags['WRITEABLE']):
        raise TypeError("%s must be a writable contiguous float64 array (it is updated in place)" % name)
    if a.size != size:
        raise ValueError("%s has %d entries, expected n_rows * n_cols = %d" % (name, a.size, size))
    return a


def drain_connections(arr, ids, indptr, indices, set_to=0, device=0):
    """cyutils.drain_connections (cyutils.pyx:35-46): flood `set_to` from the cells in `ids` along the
    CSC columns; updates and returns `arr`."""
    a, i = _u8(arr, 'arr'), _u8(ids, 'ids')
    if i.size != a.size:
        raise ValueError("ids has %d entries, arr %d" % (i.size, a.size))
    ip = _i32(indptr, 'indptr', size=a.size + 1)
    ix = _i32(indices, 'indices', at_least=int(ip[-1]))
    if ix.size and (ix.min() < 0 or ix.max() >= a.size):
        raise ValueError("indices out of range")
    if ix.size == 0:
        ix = np.zeros(1, np.int32)
    _ffi.check(_ffi.load().pydem_drain_connections(a.ctypes.data_as(_P), i.ctypes.data_as(_P), ip.ctypes.data_as(_P),
                                                  ix.ctypes.data_as(_P), a.size, int(bool(set_to)), device))
    return arr


def drain_area(area, done, ids, col_indptr, col_indices, col_data, row_indptr, row_indices, n_rows, n_cols,
               edge_todo=None, edge_todo_no_mask=None, skip_edge=0, device=0):
    """cyutils.drain_area (cyutils.pyx:78-116): level-synchronous push of area (and of the edge_todo taints)
    along the flow graph; all arrays are updated in place and returned as in the reference.  Shapes and dtypes are
    checked like the typed Cython signature does (the library reads them through raw pointers)."""
    N = int(n_rows) * int(n_cols)
    _f64_inplace(area, 'area', N)
    d, i = _u8(done, 'done'), _u8(ids, 'ids')
    if d.size != N or i.size != N:
        raise ValueError("done / ids must have n_rows * n_cols = %d entries" % N)
    cp = _i32(col_indptr, 'col_indptr', size=N + 1)
    rp = _i32(row_indptr, 'row_indptr', size=N + 1)
    nnz = int(cp[-1])
    if nnz < 0 or int(rp[-1]) != nnz:
        raise ValueError("col_indptr and row_indptr describe different numbers of edges")
    ci = _i32(col_indices, 'col_indices', at_least=nnz)
    ri = _i32(row_indices, 'row_indices', at_least=nnz)
    data = np.ascontiguousarray(col_data, np.float64)
    if data.ndim != 1 or data.size < nnz:
        raise ValueError("col_data has %d entries, expected >= %d" % (data.size, nnz))
    for nm, ix in (('col_indices', ci), ('row_indices', ri)):
        if nnz and (ix[:nnz].min() < 0 or ix[:nnz].max() >= N):
            raise ValueError("%s out of range" % nm)
    if nnz == 0:
        ci = np.zeros(1, np.int32); ri = np.zeros(1, np.int32); data = np.zeros(1)
    if edge_todo is not None:
        _f64_inplace(edge_todo, 'edge_todo', N)
    if edge_todo_no_mask is not None:
        _f64_inplace(edge_todo_no_mask, 'edge_todo_no_mask', N)
    et = None if edge_todo is None else edge_todo.ctypes.data_as(_P)
    etn = None if edge_todo_no_mask is None else edge_todo_no_mask.ctypes.data_as(_P)
    _ffi.check(_ffi.load().pydem_drain_area(area.ctypes.data_as(_P), d.ctypes.data_as(_P), i.ctypes.data_as(_P),
                                           cp.ctypes.data_as(_P), ci.ctypes.data_as(_P), data.ctypes.data_as(_P),
                                           rp.ctypes.data_as(_P), ri.ctypes.data_as(_P), int(n_rows), int(n_cols),
                                           et, etn, int(skip_edge), device))
    return area, done, edge_todo, edge_todo_no_mask
