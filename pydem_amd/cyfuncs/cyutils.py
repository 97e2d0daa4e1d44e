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


def drain_connections(arr, ids, indptr, indices, set_to=0, device=0):
    """cyutils.drain_connections (cyutils.pyx:35-46): flood `set_to` from the cells in `ids` along the
    CSC columns; updates and returns `arr`."""
    a, i = _u8(arr, 'arr'), _u8(ids, 'ids')
    ip = np.ascontiguousarray(indptr, np.int32); ix = np.ascontiguousarray(indices, np.int32)
    if ix.size == 0:
        ix = np.zeros(1, np.int32)
    _ffi.check(_ffi.load().pydem_drain_connections(a.ctypes.data_as(_P), i.ctypes.data_as(_P), ip.ctypes.data_as(_P),
                                                  ix.ctypes.data_as(_P), a.size, int(bool(set_to)), device))
    return arr


def drain_area(area, done, ids, col_indptr, col_indices, col_data, row_indptr, row_indices, n_rows, n_cols,
               edge_todo=None, edge_todo_no_mask=None, skip_edge=0, device=0):
    """cyutils.drain_area (cyutils.pyx:78-116): level-synchronous push of area (and of the edge_todo taints)
    along the flow graph; all arrays are updated in place and returned as in the reference."""
    if not (isinstance(area, np.ndarray) and area.dtype == np.float64 and area.flags['C_CONTIGUOUS']):
        raise TypeError("area must be a contiguous float64 array (it is updated in place)")
    d, i = _u8(done, 'done'), _u8(ids, 'ids')
    arrs = [np.ascontiguousarray(x, np.int32) for x in (col_indptr, col_indices, row_indptr, row_indices)]
    data = np.ascontiguousarray(col_data, np.float64)
    if arrs[1].size == 0:
        arrs[1] = np.zeros(1, np.int32); arrs[3] = np.zeros(1, np.int32); data = np.zeros(1)
    et = None if edge_todo is None else edge_todo.ctypes.data_as(_P)
    etn = None if edge_todo_no_mask is None else edge_todo_no_mask.ctypes.data_as(_P)
    _ffi.check(_ffi.load().pydem_drain_area(area.ctypes.data_as(_P), d.ctypes.data_as(_P), i.ctypes.data_as(_P),
                                           arrs[0].ctypes.data_as(_P), arrs[1].ctypes.data_as(_P), data.ctypes.data_as(_P),
                                           arrs[2].ctypes.data_as(_P), arrs[3].ctypes.data_as(_P), int(n_rows), int(n_cols),
                                           et, etn, int(skip_edge), device))
    return area, done, edge_todo, edge_todo_no_mask
