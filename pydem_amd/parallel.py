"""Multi-process tile sharding for the ProcessManager: one process per GPU, tile i lives on rank
i % world_size, and the only cross-rank traffic is the edge strips of the directory flow.

The reference shards tiles over a multiprocessing.Pool and exchanges strips through a shared
on-disk zarr store (pydem/process_manager.py:1214-1288, :243-255).  Here:

  * RcclTransport -- the production path: every rank packs the lines it owns into a device buffer
    and one ncclAllReduce(sum) over xGMI leaves all lines on all ranks (csrc/comm.hip);
  * DistTransport -- the same protocol over the socket group of `pydem_amd.rendezvous` (host memory, no framework);
    used by the CPU test tier (world_size 2) and as a fallback when RCCL cannot be initialised.

Both are driven in lock-step by the (deterministic) host logic of ProcessManager, so every rank
issues the same sequence of gathers.
"""
import os

import numpy as np

from .process_manager import EdgeTransport


def rank_world():
    return int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1'))


class DistTransport(EdgeTransport):
    """Strips through host memory: python objects gathered over a `rendezvous.SocketGroup`."""

    def __init__(self, pm, group):
        EdgeTransport.__init__(self, pm)
        self.group = group
        self.rank, self.world = group.rank, group.world

    def owns(self, i):
        return i % self.world == self.rank

    def gather_lines(self, requests):
        mine = {}
        for k, (t, name, axis, index) in enumerate(requests):
            if t >= 0 and self.owns(t):
                mine[k] = self.pm.tiles[t].get_line(name, axis, index)
        out = [None] * len(requests)
        for part in self.group.all_gather_object(mine):
            for k, v in part.items():
                out[k] = v
        return out

    def allreduce_max(self, value):
        return self.group.allreduce_max(value)

    def sum_inplace(self, arr):
        """Sum a contiguous float64 array over all ranks, in place (the edge board's staging buffer when the strips
        cannot travel by RCCL: ProcessManager._process_uca_edges_pool_device)."""
        self.group.sum_inplace(arr)

    def sum_bytes_inplace(self, arr):
        """Byte-wise sum of a uint8 array over all ranks, in place (the staging buffer of a QUEUED batch of the fix-up: the
        socket restatement of the RCCL path's ncclAllReduce(ncclUint8), ProcessManager._process_uca_edges_pool_device)."""
        self.group.sum_bytes_inplace(arr)

    def max_inplace(self, arr):
        self.group.max_inplace(arr)

    def barrier(self):
        self.group.barrier()


class RcclTransport(EdgeTransport):
    """Strips over RCCL/xGMI: device pack -> one all-reduce -> host views."""

    _BOOL = ('flats', 'edge_todo', 'edge_done')

    def __init__(self, pm, comm):
        EdgeTransport.__init__(self, pm)
        self.comm = comm
        self.rank, self.world = comm.rank, comm.world

    def owns(self, i):
        return i % self.world == self.rank

    def gather_lines(self, requests):
        from .dem_processing import _FIELD_OF
        lens, offs, total = [], [], 0
        for t, name, axis, index in requests:
            n = 0 if t < 0 else int(self.pm.index[t, 7] if axis == 0 else self.pm.index[t, 6])
            lens.append(n); offs.append(total); total += n
        if total == 0:
            return [None] * len(requests)
        self.comm.begin(total)
        mine = {}
        for (t, name, axis, index), off in zip(requests, offs):
            if t >= 0 and self.owns(t):
                mine.setdefault(t, []).append((name, axis, index, off))
        for t, lines in mine.items():          # one launch sequence per tile, ordered against the collective by stream events
            dp = self.pm.tiles[t]
            dp._ensure_tile()
            dp._push(*sorted(set(l[0] for l in lines)))
            self.comm.pack_lines(dp._tile, [(_FIELD_OF[name], axis, index, off) for name, axis, index, off in lines])
        flat = self.comm.allreduce(total, op=0)
        out = []
        for (t, name, axis, index), off, n in zip(requests, offs, lens):
            if t < 0:
                out.append(None)
                continue
            v = flat[off:off + n].copy()
            out.append(v != 0 if name in self._BOOL else v)
        return out

    def allreduce_max(self, value):
        self.comm.begin(1)
        self.comm.put([float(value)])
        return float(self.comm.allreduce(1, op=1)[0])

    def barrier(self):
        self.allreduce_max(0.0)


def make_rccl_transport(pm, device, group):
    """Create the RCCL communicator for this rank.  The 128-byte id travels from rank 0 over the socket group
    (plumbing only)."""
    from . import _ffi
    uid = group.broadcast_object(_ffi.Comm.unique_id() if group.rank == 0 else None, src=0)
    comm = _ffi.Comm(group.world, group.rank, uid, device)
    return RcclTransport(pm, comm)
